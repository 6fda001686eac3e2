#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference planners (build container only).

Test infrastructure: imports the reference classes through oracle/ref_loader.py
(definitions only, no driver code), seeds CPython's `random` explicitly, runs
`planning(animation=False)` and stores inputs + outputs as .npz under
tests/golden/.  Only data is stored (final SoA tree, returned path, a compact
per-iteration trace); no reference source text enters the repo.

Environment of the goldens (part of the fixture): CPython 3.10.12,
numpy 2.2.6, glibc 2.35 (x86-64 FMA ifunc variants), this container's Xeon.

Usage: python oracle/gen_golden.py [--only PREFIX] [--big]
"""
import argparse
import contextlib
import io
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_loader  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

DRIVER_OBST = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)]


def synth_map(map_seed, m, rmin=0.5, rmax=2.5):
    """SURVEY.md 8(d) / section 10 map generator (private Random, independent of the planner stream)."""
    rng = random.Random(map_seed)
    obs = []
    while len(obs) < m:
        x = rng.uniform(0, 100)
        y = rng.uniform(0, 100)
        r = rng.uniform(rmin, rmax)
        ok = True
        for (kx, ky) in ((2, 2), (98, 98)):
            if not ((x - kx) ** 2 + (y - ky) ** 2 > (r + 3) ** 2):
                ok = False
        if ok:
            obs.append((x, y, r))
    return obs


def tree_arrays(node_list, int_parent=False):
    n = len(node_list)
    x = np.array([float(nd.x) for nd in node_list], dtype=np.float64)
    y = np.array([float(nd.y) for nd in node_list], dtype=np.float64)
    cost = np.array([float(nd.cost) for nd in node_list], dtype=np.float64) if hasattr(node_list[0], "cost") \
        else np.zeros(n)
    if int_parent:
        parent = np.array([-1 if nd.parent is None else int(nd.parent) for nd in node_list], dtype=np.int32)
    else:
        ids = {id(nd): i for i, nd in enumerate(node_list)}
        parent = np.array([-1 if nd.parent is None else ids.get(id(nd.parent), -2) for nd in node_list],
                          dtype=np.int32)
    return x, y, cost, parent


def run_rrt04(mod, name, obstacles, start, goal, rand_area, expand_dis, path_resolution, goal_sample_rate, max_iter,
              play_area, robot_radius, sobol, ccd, until_max, seed, trace=True, algo="rrt_star"):
    ref_loader.reset_sobol(mod)
    random.seed(seed)
    kw = dict(start=start, goal=goal, obstacle_list=obstacles, rand_area=rand_area, expand_dis=expand_dis,
              path_resolution=path_resolution, goal_sample_rate=goal_sample_rate, max_iter=max_iter,
              play_area=play_area, robot_radius=robot_radius)
    if algo == "rrt_star":
        kw.update(sobol_sampler=sobol, connect_circle_dist=ccd, search_until_max_iter=until_max)
    if algo == "rrt" and sobol:
        pass  # rrt_02: the class itself always draws Sobol points
    rrt = mod.RRT(**kw)
    tr = {"rnd_x": [], "rnd_y": [], "nearest": [], "n_near": [], "n_nodes": []}
    edges = [0]
    if trace:
        cls = mod.RRT
        o_near = cls.get_nearest_node_index
        o_cc = cls.check_collision

        def near_hook(node_list, rnd):
            i = o_near(node_list, rnd)
            tr["rnd_x"].append(float(rnd.x))
            tr["rnd_y"].append(float(rnd.y))
            tr["nearest"].append(i)
            tr["n_nodes"].append(len(node_list))
            return i

        def cc_hook(node, ol, rr):
            edges[0] += 1
            return o_cc(node, ol, rr)
        rrt.get_nearest_node_index = near_hook
        rrt.check_collision = cc_hook
        if algo == "rrt_star":
            o_fn = rrt.find_near_nodes

            def fn_hook(new_node):
                r = o_fn(new_node)
                tr["n_near"].append(len(r))
                return r
            rrt.find_near_nodes = fn_hook
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        path = rrt.planning(animation=False)
    dt = time.time() - t0
    x, y, cost, parent = tree_arrays(rrt.node_list)
    assert (parent != -2).all(), "stale parent object"
    state = random.getstate()
    out = dict(
        algo=algo, seed=seed, obstacles=np.array(obstacles, dtype=np.float64), start=np.array(start, dtype=np.float64),
        goal=np.array(goal, dtype=np.float64), rand_area=np.array(rand_area, dtype=np.float64),
        expand_dis=expand_dis, path_resolution=path_resolution, goal_sample_rate=goal_sample_rate,
        max_iter=max_iter, play_area=np.array(play_area if play_area is not None else [], dtype=np.float64),
        robot_radius=robot_radius, sobol=int(bool(sobol)), connect_circle_dist=ccd, until_max=int(bool(until_max)),
        x=x, y=y, cost=cost, parent=parent,
        path=np.array(path if path is not None else [], dtype=np.float64).reshape(-1, 2),
        path_found=int(path is not None), ref_seconds=dt, ref_edges=edges[0],
        rng_pos_after=state[1][624], rng_word0_after=np.uint32(state[1][0]),
        sobol_index_after=getattr(rrt, "sobol_inter_", 0),
        tr_rnd_x=np.array(tr["rnd_x"]), tr_rnd_y=np.array(tr["rnd_y"]),
        tr_nearest=np.array(tr["nearest"], dtype=np.int32), tr_n_near=np.array(tr["n_near"], dtype=np.int32),
        tr_n_nodes=np.array(tr["n_nodes"], dtype=np.int32),
    )
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-28s nodes=%d path=%s edges=%d  %.2fs" % (name, len(x), None if path is None else len(path), edges[0], dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--big", action="store_true", help="also the 8000-iteration C2 case (~80 s)")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)

    def want(n):
        return n.startswith(a.only)

    m04 = ref_loader.load("rrt_04")
    m01 = ref_loader.load("rrt_01")
    drv = dict(obstacles=DRIVER_OBST, start=[0, 0], goal=[6.0, 10.0], rand_area=[-2, 15], expand_dis=1.0,
               path_resolution=0.1, goal_sample_rate=5, max_iter=500, play_area=[0, 10, 0, 14], robot_radius=0.6,
               ccd=50.0)
    # rrt_04 driver scenario (rrt_04:1498-1546)
    for seed in (1234, 5, 42):
        for sob in (0, 1):
            n = "rrt04_drv_%s_s%d" % ("sobol" if sob else "mt", seed)
            if want(n):
                run_rrt04(m04, n, sobol=sob, until_max=True, seed=seed, **drv)
    for seed in (1, 2, 3, 7):
        n = "rrt04_drv_early_s%d" % seed
        if want(n):
            run_rrt04(m04, n, sobol=0, until_max=False, seed=seed, **drv)
    n = "rrt04_drv_noplay_s9"
    if want(n):
        d2 = dict(drv)
        d2["play_area"] = None
        d2["max_iter"] = 1500
        run_rrt04(m04, n, sobol=0, until_max=True, seed=9, **d2)
    n = "rrt04_drv_long_s5"
    if want(n):
        d2 = dict(drv)
        d2["max_iter"] = 1500
        run_rrt04(m04, n, sobol=0, until_max=True, seed=5, **d2)
    # C2-style synthetic map
    c2 = dict(obstacles=synth_map(7, 50), start=[2, 2], goal=[98, 98], rand_area=[0, 100], expand_dis=2.0,
              path_resolution=0.25, goal_sample_rate=5, play_area=None, robot_radius=0.0, ccd=50.0)
    for it in (1000, 2000, 4000) + ((8000,) if a.big else ()):
        n = "rrt04_c2_s1_it%d" % it
        if want(n):
            run_rrt04(m04, n, sobol=0, until_max=True, seed=1, max_iter=it, **c2)
    for seed in (2, 3):
        n = "rrt04_c2_s%d_it1500" % seed
        if want(n):
            run_rrt04(m04, n, sobol=0, until_max=True, seed=seed, max_iter=1500, **c2)
    n = "rrt04_c2_sobol_s4_it1500"
    if want(n):
        run_rrt04(m04, n, sobol=1, until_max=True, seed=4, max_iter=1500, **c2)
    # edge cases (empty / degenerate inputs), the reference run on each: driver scenario with one thing changed
    edge = dict(drv)
    edge["max_iter"] = 300
    for tag, upd in (("noobst", dict(obstacles=[])),                       # empty obstacle list
                     ("iter0", dict(max_iter=0)), ("iter1", dict(max_iter=1)),
                     ("startblocked", dict(start=[5.0, 5.0])),               # start inside a circle: nothing ever extends
                     ("goalinside", dict(goal=[3.0, 8.0])),                  # goal inside a circle: no path
                     ("rate100", dict(goal_sample_rate=100)), ("rate0", dict(goal_sample_rate=0)),
                     ("bigstep", dict(expand_dis=30.0)),                     # every extension reaches its sample
                     ("substep", dict(expand_dis=0.05)),                     # expand_dis < path_resolution: floor() = 0
                     ("startgoal", dict(goal=[0.0, 0.0])),                   # start == goal
                     ("revarea", dict(rand_area=[15, -2])),                  # random.uniform(a, b) with a > b
                     ("bigrobot", dict(robot_radius=2.5, play_area=None)),
                     ("tinyplay", dict(play_area=[-0.5, 0.5, -0.5, 0.5]))):  # play area around the start only
        for until in (1, 0):
            n = "rrt04_edge_%s_%s" % (tag, "full" if until else "early")
            if want(n):
                d2 = dict(edge)
                d2.update(upd)
                run_rrt04(m04, n, sobol=0, until_max=bool(until), seed=3, **d2)
        n = "rrt01_edge_%s" % tag
        if want(n) and tag != "tinyplay":
            d2 = dict(edge)
            d2.update(upd)
            d2["play_area"] = None
            run_rrt04(m01, n, sobol=0, until_max=False, seed=3, algo="rrt", **d2)
    # rrt_01 driver scenario (rrt_01:354-391), seeds 0..15 (config C1)
    d1 = dict(drv)
    d1["play_area"] = None
    for seed in list(range(16)) + [42]:
        n = "rrt01_drv_s%d" % seed
        if want(n):
            run_rrt04(m01, n, sobol=0, until_max=False, seed=seed, algo="rrt", **d1)


if __name__ == "__main__":
    main()


# ------------------------------------------------------------------------------------------------ rrt_07
def run_rrt07(mod, name, obstacles, start, goal, rand_area, expand_dis, goal_sample_rate, max_iter, sobol, seed):
    """Informed RRT* (rrt_07:1027-1285).  Also stores the rotation matrix C the reference builds with numpy's SVD
    (rrt_07:1061-1068): the product takes it from the host exactly like this."""
    import math
    ref_loader.reset_sobol(mod)
    random.seed(seed)
    rrt = mod.RRT(start=start, goal=goal, obstacle_list=obstacles, rand_area=rand_area, expand_dis=expand_dis,
                  goal_sample_rate=goal_sample_rate, max_iter=max_iter, sobol_sampler=sobol)
    tr = {"rnd_x": [], "rnd_y": [], "nearest": [], "n_near": []}
    cm = {}
    o_is = rrt.informed_sample
    o_near = mod.RRT.get_nearest_list_index
    o_fn = rrt.find_near_nodes

    def is_hook(c_max, c_min, x_center, c):
        cm["c"] = np.array(c, dtype=np.float64)
        cm["c_min"] = float(c_min)
        cm["x_center"] = np.array(x_center, dtype=np.float64)
        r = o_is(c_max, c_min, x_center, c)
        tr["rnd_x"].append(float(r[0]))
        tr["rnd_y"].append(float(r[1]))
        return r

    def near_hook(nodes, rnd):
        i = o_near(nodes, rnd)
        tr["nearest"].append(i)
        return i

    def fn_hook(new_node):
        r = o_fn(new_node)
        tr["n_near"].append(len(r))
        return r
    rrt.informed_sample = is_hook
    rrt.get_nearest_list_index = near_hook
    rrt.find_near_nodes = fn_hook
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        path = rrt.informed_rrt_star_search(animation=False)
    dt = time.time() - t0
    x, y, cost, parent = tree_arrays(rrt.node_list, int_parent=True)
    state = random.getstate()
    if "c" not in cm:   # max_iter = 0: the loop never samples, so the hook never saw the ellipse terms (unused in that
        import oracle    # run); stored from the numpy restatement that the other rrt_07 goldens pin
        cm.update(c=oracle.rotation_to_world(list(start), list(goal)),
                  c_min=math.hypot(start[0] - goal[0], start[1] - goal[1]),
                  x_center=np.array([[(start[0] + goal[0]) / 2.0], [(start[1] + goal[1]) / 2.0], [0]]))
    out = dict(
        algo="informed", seed=seed, obstacles=np.array(obstacles, dtype=np.float64),
        start=np.array(start, dtype=np.float64), goal=np.array(goal, dtype=np.float64),
        rand_area=np.array(rand_area, dtype=np.float64), expand_dis=expand_dis, goal_sample_rate=goal_sample_rate,
        max_iter=max_iter, sobol=int(bool(sobol)), rot_c=cm["c"], c_min=cm["c_min"], x_center=cm["x_center"],
        x=x, y=y, cost=cost, parent=parent,
        path=np.array(path if path is not None else [], dtype=np.float64).reshape(-1, 2),
        path_found=int(path is not None), ref_seconds=dt,
        path_len=float(mod.RRT.get_path_len(path)) if path is not None else float("inf"),
        rng_pos_after=state[1][624], rng_word0_after=np.uint32(state[1][0]),
        sobol_index_after=getattr(rrt, "sobol_inter_", 0),
        tr_rnd_x=np.array(tr["rnd_x"]), tr_rnd_y=np.array(tr["rnd_y"]),
        tr_nearest=np.array(tr["nearest"], dtype=np.int32), tr_n_near=np.array(tr["n_near"], dtype=np.int32),
    )
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-28s nodes=%d path=%s len=%r  %.2fs" % (name, len(x), None if path is None else len(path), out["path_len"], dt))


def main07(only=""):
    os.makedirs(OUT, exist_ok=True)
    m07 = ref_loader.load("rrt_07")
    drv = dict(obstacles=DRIVER_OBST, start=[0.0, 0.0], goal=[6.0, 10.0], rand_area=[-2, 15], expand_dis=0.5,
               goal_sample_rate=10)
    jobs = []
    for seed, it, sob in ((42, 200, 0), (42, 200, 1), (42, 2000, 0), (7, 1200, 0), (3, 1200, 1), (11, 800, 0), (5, 600, 0)):
        jobs.append(("rrt07_drv_%s_s%d_it%d" % ("sobol" if sob else "mt", seed, it), dict(drv, max_iter=it, sobol=sob, seed=seed)))
    # C3-style synthetic map: map_seed 11, 200 circles, radii U(0.3,1.5), Sobol sampler
    c3 = dict(obstacles=synth_map(11, 200, 0.3, 1.5), start=[2, 2], goal=[98, 98], rand_area=[0, 100], expand_dis=0.5,
              goal_sample_rate=10)
    for seed, it, sob in ((1, 3000, 1), (2, 3000, 0)):
        jobs.append(("rrt07_c3_%s_s%d_it%d" % ("sobol" if sob else "mt", seed, it), dict(c3, max_iter=it, sobol=sob, seed=seed)))
    # a denser, closer problem on the C3 obstacle family so that a path (and the informed phase) is reached quickly
    c3b = dict(obstacles=synth_map(11, 200, 0.3, 1.5), start=[40, 40], goal=[50, 52], rand_area=[30, 62], expand_dis=0.5,
               goal_sample_rate=10)
    for seed, it, sob in ((3, 700, 1), (4, 700, 0)):
        jobs.append(("rrt07_c3near_%s_s%d_it%d" % ("sobol" if sob else "mt", seed, it), dict(c3b, max_iter=it, sobol=sob, seed=seed)))
    # edge cases: the driver scenario with one thing changed
    for tag, upd in (("noobst", dict(obstacles=[])), ("iter0", dict(max_iter=0)), ("iter1", dict(max_iter=1)),
                     ("startblocked", dict(start=[5.0, 5.0])), ("goalinside", dict(goal=[3.0, 8.0])),
                     ("rate100", dict(goal_sample_rate=100))):   # start == goal: the reference divides by c_min = 0 (:1058)
        jobs.append(("rrt07_edge_%s" % tag, dict(dict(drv, max_iter=300, sobol=0, seed=3), **upd)))
    for n, kw in jobs:
        if n.startswith(only):
            run_rrt07(m07, n, **kw)


def main02():
    """rrt_02 (plain RRT + Sobol sampler) driver scenario, rrt_02:1290-1330 constants = rrt_01's."""
    m02 = ref_loader.load("rrt_02")
    drv = dict(obstacles=DRIVER_OBST, start=[0, 0], goal=[6.0, 10.0], rand_area=[-2, 15], expand_dis=1.0,
               path_resolution=0.1, goal_sample_rate=5, max_iter=500, play_area=None, robot_radius=0.6, ccd=50.0)
    for seed in (0, 1, 2, 3, 42):
        run_rrt04(m02, "rrt02_drv_s%d" % seed, sobol=1, until_max=False, seed=seed, algo="rrt", **drv)


# ------------------------------------------------------------------------------------------------ rrt_05
def run_rrt05(mod, name, obstacles, start, goal, rand_area, max_iter, seed, curvature=1.0, robot_radius=0.0,
              goal_sample_rate=10, expand_dis=3.0, ccd=50.0, until_max=True):
    """RRT*-Dubins (rrt_05:1335-1795), driver-style call planning(animation=False)."""
    random.seed(seed)
    rrt = mod.RRT(start=start, goal=goal, obstacle_list=obstacles, rand_area=rand_area, expand_dis=expand_dis,
                  path_resolution=0.5, goal_sample_rate=goal_sample_rate, max_iter=max_iter, play_area=None,
                  robot_radius=robot_radius, sobol_sampler=True, connect_circle_dist=ccd, search_until_max_iter=True,
                  curvature=curvature, goal_yaw_th=np.deg2rad(1.0), goal_xy_th=0.5)
    tr = {"rx": [], "ry": [], "ryaw": [], "nearest": [], "n_near": []}
    o_near = mod.RRT.get_nearest_node_index
    o_fn = rrt.find_near_nodes

    def near_hook(node_list, rnd):
        i = o_near(node_list, rnd)
        tr["rx"].append(float(rnd.x)); tr["ry"].append(float(rnd.y)); tr["ryaw"].append(float(rnd.yaw))
        tr["nearest"].append(i)
        return i

    def fn_hook(new_node):
        r = o_fn(new_node)
        tr["n_near"].append(len(r))
        return r
    rrt.get_nearest_node_index = near_hook
    rrt.find_near_nodes = fn_hook
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        path = rrt.planning(animation=False) if until_max else rrt.planning(animation=False, search_until_max_iter=False)
    dt = time.time() - t0
    nl = rrt.node_list
    x, y, cost, parent = tree_arrays(nl)
    yaw = np.array([float(nd.yaw) for nd in nl])
    plen = np.array([len(nd.path_x) for nd in nl], dtype=np.int32)
    ppx = np.concatenate([np.asarray(nd.path_x, dtype=np.float64).reshape(-1) for nd in nl]) if len(nl) else np.zeros(0)
    ppy = np.concatenate([np.asarray(nd.path_y, dtype=np.float64).reshape(-1) for nd in nl]) if len(nl) else np.zeros(0)
    state = random.getstate()
    out = dict(algo="rrt_star_dubins", seed=seed, search_until_max_iter=int(bool(until_max)), obstacles=np.array(obstacles, dtype=np.float64),
               start=np.array(start, dtype=np.float64), goal=np.array(goal, dtype=np.float64),
               rand_area=np.array(rand_area, dtype=np.float64), max_iter=max_iter, curvature=curvature,
               robot_radius=robot_radius, goal_sample_rate=goal_sample_rate, expand_dis=expand_dis,
               connect_circle_dist=ccd, goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5,
               x=x, y=y, yaw=yaw, cost=cost, parent=parent, poly_len=plen, poly_x=ppx, poly_y=ppy,
               path=np.array(path if path is not None else [], dtype=np.float64).reshape(-1, 2),
               path_found=int(path is not None), ref_seconds=dt,
               rng_pos_after=state[1][624], rng_word0_after=np.uint32(state[1][0]),
               tr_rx=np.array(tr["rx"]), tr_ry=np.array(tr["ry"]), tr_ryaw=np.array(tr["ryaw"]),
               tr_nearest=np.array(tr["nearest"], dtype=np.int32), tr_n_near=np.array(tr["n_near"], dtype=np.int32))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-28s nodes=%d path=%s maxcost=%r  %.2fs" % (name, len(x), None if path is None else len(path),
                                                        float(cost.max()), dt), flush=True)


def main05(only=""):
    m05 = ref_loader.load("rrt_05")
    obst = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2)]       # rrt_05:1804-1806
    drv = dict(obstacles=obst, start=[0.0, 0.0, float(np.deg2rad(0.0))], goal=[10.0, 10.0, float(np.deg2rad(0.0))],
               rand_area=[-2, 15])
    for seed, it in ((42, 150), (42, 500), (1, 300), (2, 300), (7, 400), (374, 3000)):   # 374: a node rewired twice in one iteration
        n = "rrt05_drv_s%d_it%d" % (seed, it)
        if n.startswith(only):
            run_rrt05(m05, n, max_iter=it, seed=seed, **drv)
    for tag, upd in (("noobst", dict(obstacles=[])), ("iter0", dict(max_iter=0)), ("iter1", dict(max_iter=1)),
                     ("startblocked", dict(start=[5.0, 5.0, 0.0])), ("goalinside", dict(goal=[3.0, 8.0, 0.0])),
                     ("startgoal", dict(goal=[0.0, 0.0, 0.0]))):
        n = "rrt05_edge_%s" % tag
        if n.startswith(only):
            run_rrt05(m05, n, **dict(dict(drv, max_iter=200, seed=3), **upd))
    if only.startswith("rrt05_edge") or only.startswith("rrt05_drv"):
        return
    # Dubins primitive known-answer vectors: plan_dubins_path on random poses (no planner around it)
    rng = random.Random(5)
    rows = []
    for k in range(400):
        sx, sy, syaw = rng.uniform(-2, 15), rng.uniform(-2, 15), rng.uniform(-math_pi(), math_pi())
        gx, gy, gyaw = rng.uniform(-2, 15), rng.uniform(-2, 15), rng.uniform(-math_pi(), math_pi())
        if k % 7 == 0:
            gx, gy = sx + rng.uniform(-0.3, 0.3), sy + rng.uniform(-0.3, 0.3)
        px, py, pyaw, mode, lengths = m05.plan_dubins_path(sx, sy, syaw, gx, gy, gyaw, 1.0)
        rows.append(dict(inp=[sx, sy, syaw, gx, gy, gyaw], n=len(px), end=[float(px[-1]), float(py[-1]), float(pyaw[-1])],
                         lengths=[float(v) for v in lengths], mode="".join(mode),
                         chk=[float(np.sum(px)), float(np.sum(py))], px=np.asarray(px, dtype=np.float64),
                         py=np.asarray(py, dtype=np.float64)))
    np.savez_compressed(os.path.join(OUT, "dubins_kat.npz"),
                        inp=np.array([r["inp"] for r in rows]), n=np.array([r["n"] for r in rows], dtype=np.int32),
                        end=np.array([r["end"] for r in rows]), lengths=np.array([r["lengths"] for r in rows]),
                        mode=np.array([r["mode"] for r in rows]),
                        poly_x=np.concatenate([r["px"] for r in rows]), poly_y=np.concatenate([r["py"] for r in rows]))
    print("dubins_kat: 400 cases", flush=True)


# ------------------------------------------------------------------------------------------------ rrt_03
def run_rrt03(mod, name, obstacles, start, goal, rand_area, max_iter, seed, sobol, curvature=1.0, robot_radius=0.6,
              goal_sample_rate=10, until_max=True):
    """RRT with Dubins steer (rrt_03:1348-1700), driver-style call planning(animation=False)."""
    ref_loader.reset_sobol(mod)
    random.seed(seed)
    rrt = mod.RRT(start=start, goal=goal, obstacle_list=obstacles, rand_area=rand_area,
                  goal_sample_rate=goal_sample_rate, max_iter=max_iter, play_area=None, robot_radius=robot_radius,
                  sobol_sampler=bool(sobol), curvature=curvature, goal_yaw_th=np.deg2rad(1.0), goal_xy_th=0.5)
    tr = {"rx": [], "ry": [], "ryaw": [], "nearest": []}
    o_near = mod.RRT.get_nearest_node_index

    def near_hook(node_list, rnd):
        i = o_near(node_list, rnd)
        tr["rx"].append(float(rnd.x)); tr["ry"].append(float(rnd.y)); tr["ryaw"].append(float(rnd.yaw))
        tr["nearest"].append(i)
        return i
    rrt.get_nearest_node_index = near_hook
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        path = rrt.planning(animation=False) if until_max else rrt.planning(animation=False, search_until_max_iter=False)
    dt = time.time() - t0
    nl = rrt.node_list
    x, y, cost, parent = tree_arrays(nl)
    yaw = np.array([float(nd.yaw) for nd in nl])
    plen = np.array([len(nd.path_x) for nd in nl], dtype=np.int32)
    ppx = np.concatenate([np.asarray(nd.path_x, dtype=np.float64).reshape(-1) for nd in nl]) if len(nl) else np.zeros(0)
    ppy = np.concatenate([np.asarray(nd.path_y, dtype=np.float64).reshape(-1) for nd in nl]) if len(nl) else np.zeros(0)
    state = random.getstate()
    out = dict(algo="rrt_dubins", seed=seed, search_until_max_iter=int(bool(until_max)), sobol=int(bool(sobol)), obstacles=np.array(obstacles, dtype=np.float64),
               start=np.array(start, dtype=np.float64), goal=np.array(goal, dtype=np.float64),
               rand_area=np.array(rand_area, dtype=np.float64), max_iter=max_iter, curvature=curvature,
               robot_radius=robot_radius, goal_sample_rate=goal_sample_rate, goal_yaw_th=float(np.deg2rad(1.0)),
               goal_xy_th=0.5, x=x, y=y, yaw=yaw, cost=cost, parent=parent, poly_len=plen, poly_x=ppx, poly_y=ppy,
               path=np.array(path if path is not None else [], dtype=np.float64).reshape(-1, 2),
               path_found=int(path is not None), ref_seconds=dt, sobol_index_after=int(rrt.sobol_inter_),
               rng_pos_after=state[1][624], rng_word0_after=np.uint32(state[1][0]),
               tr_rx=np.array(tr["rx"]), tr_ry=np.array(tr["ry"]), tr_ryaw=np.array(tr["ryaw"]),
               tr_nearest=np.array(tr["nearest"], dtype=np.int32))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-28s nodes=%d path=%s maxcost=%r  %.2fs" % (name, len(x), None if path is None else len(path),
                                                        float(cost.max()), dt), flush=True)


def main03(only=""):
    m03 = ref_loader.load("rrt_03")
    obst = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2)]       # rrt_03 driver
    drv = dict(obstacles=obst, start=[0.0, 0.0, float(np.deg2rad(0.0))], goal=[10.0, 10.0, float(np.deg2rad(0.0))],
               rand_area=[-2, 15])
    for seed, it, sob in ((42, 200, 1), (1, 200, 1), (2, 400, 1), (3, 1000, 1), (42, 200, 0), (5, 600, 0), (9, 1500, 0),
                          (11, 3000, 1)):
        n = "rrt03_drv_s%d_it%d_%s" % (seed, it, "sobol" if sob else "mt")
        if n.startswith(only):
            run_rrt03(m03, n, max_iter=it, seed=seed, sobol=sob, **drv)


def main_early(only=""):
    """planning(animation=False, search_until_max_iter=False): the early-return mode of rrt_03 / rrt_05 (:1443-1446)."""
    obst = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2)]
    drv = dict(obstacles=obst, start=[0.0, 0.0, float(np.deg2rad(0.0))], goal=[10.0, 10.0, float(np.deg2rad(0.0))],
               rand_area=[-2, 15])
    m03 = ref_loader.load("rrt_03")
    for seed, it, sob in ((42, 1500, 1), (1, 1500, 0), (5, 1500, 0), (8, 60, 1)):
        n = "rrt03_early_s%d_it%d_%s" % (seed, it, "sobol" if sob else "mt")
        if n.startswith(only):
            run_rrt03(m03, n, max_iter=it, seed=seed, sobol=sob, until_max=False, **drv)
    m05 = ref_loader.load("rrt_05")
    for seed, it in ((42, 2000), (3, 1500), (9, 1500), (4, 100)):
        n = "rrt05_early_s%d_it%d" % (seed, it)
        if n.startswith(only):
            run_rrt05(m05, n, max_iter=it, seed=seed, until_max=False, **drv)


def main_smooth(only=""):
    """path_smoothing (rrt_04:1447-1479) as the driver uses it (:1548-1559): random.seed -> planning() ->
    path_smoothing(path, 1000, obstacleList) on the stream planning() left behind; plus stand-alone polylines."""
    m04 = ref_loader.load("rrt_04")

    def smooth_and_store(name, path, iters, obst):
        st0 = random.getstate()
        t0 = time.time()
        sm = m04.path_smoothing([list(p) for p in path], iters, obst)
        dt = time.time() - t0
        st1 = random.getstate()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), algo="path_smoothing",
                            path_in=np.array(path, dtype=np.float64), smoothed=np.array(sm, dtype=np.float64),
                            max_iter=iters, obstacles=np.array(obst, dtype=np.float64),
                            rng_mt_before=np.array(st0[1][:624], dtype=np.uint32), rng_pos_before=st0[1][624],
                            rng_pos_after=st1[1][624], rng_word0_after=np.uint32(st1[1][0]),
                            len_in=float(m04.get_path_length(path)), len_out=float(m04.get_path_length(sm)))
        print("%-28s in=%d out=%d len %.4f -> %.4f  %.2fs" % (name, len(path), len(sm), m04.get_path_length(path),
                                                               m04.get_path_length(sm), dt), flush=True)

    drv = dict(obstacles=DRIVER_OBST, start=[0, 0], goal=[6.0, 10.0], rand_area=[-2, 15], expand_dis=1.0,
               path_resolution=0.1, goal_sample_rate=5, max_iter=500, play_area=[0, 10, 0, 14], robot_radius=0.6)
    for seed in (1234, 5, 42, 7):
        name = "smooth_drv_s%d" % seed
        if not name.startswith(only):
            continue
        ref_loader.reset_sobol(m04)
        random.seed(seed)
        rrt = m04.RRT(start=drv["start"], goal=drv["goal"], obstacle_list=drv["obstacles"], rand_area=drv["rand_area"],
                      expand_dis=drv["expand_dis"], path_resolution=drv["path_resolution"],
                      goal_sample_rate=drv["goal_sample_rate"], max_iter=drv["max_iter"], play_area=drv["play_area"],
                      robot_radius=drv["robot_radius"], sobol_sampler=False, connect_circle_dist=50.0,
                      search_until_max_iter=True)
        with contextlib.redirect_stdout(io.StringIO()):
            path = rrt.planning(animation=False)
        if path is not None:
            smooth_and_store(name, path, 1000, drv["obstacles"])
    # C2-style map, longer path
    c2 = dict(obstacles=synth_map(7, 50), start=[2.0, 2.0], goal=[98.0, 98.0], rand_area=[0, 100])
    for seed in (1, 3):
        name = "smooth_c2_s%d" % seed
        if not name.startswith(only):
            continue
        ref_loader.reset_sobol(m04)
        random.seed(seed)
        rrt = m04.RRT(start=c2["start"], goal=c2["goal"], obstacle_list=c2["obstacles"], rand_area=c2["rand_area"],
                      expand_dis=2.0, path_resolution=0.25, goal_sample_rate=5, max_iter=2500, play_area=None,
                      robot_radius=0.0, sobol_sampler=False, connect_circle_dist=50.0, search_until_max_iter=True)
        with contextlib.redirect_stdout(io.StringIO()):
            path = rrt.planning(animation=False)
        if path is not None:
            smooth_and_store(name, path, 1000, c2["obstacles"])
    # stand-alone polylines (no planner): a staircase through free space, own seed
    for seed, npts in ((11, 40), (12, 150)):
        name = "smooth_poly_s%d" % seed
        if not name.startswith(only):
            continue
        rng = random.Random(100 + seed)
        pts = [[0.0, 0.0]]
        for k in range(npts):
            pts.append([pts[-1][0] + rng.uniform(0.2, 1.0), pts[-1][1] + rng.uniform(-0.8, 1.0)])
        obst = [(pts[npts // 3][0] + 3.0, pts[npts // 3][1] - 3.0, 1.0), (pts[2 * npts // 3][0] - 3.0, pts[2 * npts // 3][1] + 4.0, 1.5)]
        random.seed(seed)
        smooth_and_store(name, pts[::-1], 600, obst)


def run_rrt06(mod, name, obstacles, start, goal, rand_area, max_iter, seed, curvature=2.0, robot_radius=0.6,
              expand_dis=3.0, ccd=50.0, step_size=0.1, until_max=True):
    """RRT*-Reeds-Shepp (rrt_06:1444-1914), driver-style call planning(animation=False) (:2087: search_until_max_iter
    is planning()'s own default True; the constructor's flag is not read)."""
    random.seed(seed)
    rrt = mod.RRT(start=start, goal=goal, obstacle_list=obstacles, rand_area=rand_area, expand_dis=expand_dis,
                  path_resolution=0.5, goal_sample_rate=10, max_iter=max_iter, play_area=None,
                  robot_radius=robot_radius, sobol_sampler=True, connect_circle_dist=ccd, search_until_max_iter=False,
                  curvature=curvature, goal_yaw_th=np.deg2rad(1.0), goal_xy_th=0.5, step_size=step_size)
    tr = {"rx": [], "ry": [], "ryaw": [], "nearest": [], "n_near": []}
    o_near = mod.RRT.get_nearest_node_index
    o_fn = rrt.find_near_nodes

    def near_hook(node_list, rnd):
        i = o_near(node_list, rnd)
        tr["rx"].append(float(rnd.x)); tr["ry"].append(float(rnd.y)); tr["ryaw"].append(float(rnd.yaw))
        tr["nearest"].append(i)
        return i

    def fn_hook(new_node):
        r = o_fn(new_node)
        tr["n_near"].append(len(r))
        return r
    rrt.get_nearest_node_index = near_hook
    rrt.find_near_nodes = fn_hook
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        path = rrt.planning(animation=False) if until_max else rrt.planning(animation=False, search_until_max_iter=False)
    dt = time.time() - t0
    nl = rrt.node_list
    x, y, cost, parent = tree_arrays(nl)
    yaw = np.array([float(nd.yaw) for nd in nl])
    plen = np.array([len(nd.path_x) for nd in nl], dtype=np.int32)
    ppx = np.concatenate([np.asarray(nd.path_x, dtype=np.float64).reshape(-1) for nd in nl])
    ppy = np.concatenate([np.asarray(nd.path_y, dtype=np.float64).reshape(-1) for nd in nl])
    state = random.getstate()
    out = dict(algo="rrt_star_reeds_shepp", seed=seed, search_until_max_iter=int(bool(until_max)),
               obstacles=np.array(obstacles, dtype=np.float64), start=np.array(start, dtype=np.float64),
               goal=np.array(goal, dtype=np.float64), rand_area=np.array(rand_area, dtype=np.float64), max_iter=max_iter,
               curvature=curvature, robot_radius=robot_radius, expand_dis=expand_dis, connect_circle_dist=ccd,
               step_size=step_size, goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5,
               x=x, y=y, yaw=yaw, cost=cost, parent=parent, poly_len=plen, poly_x=ppx, poly_y=ppy,
               path=np.array(path if path is not None else [], dtype=np.float64).reshape(-1, 3),
               path_found=int(path is not None), ref_seconds=dt,
               rng_pos_after=state[1][624], rng_word0_after=np.uint32(state[1][0]),
               tr_rx=np.array(tr["rx"]), tr_ry=np.array(tr["ry"]), tr_ryaw=np.array(tr["ryaw"]),
               tr_nearest=np.array(tr["nearest"], dtype=np.int32), tr_n_near=np.array(tr["n_near"], dtype=np.int32))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-28s nodes=%d path=%s maxcost=%r  %.2fs" % (name, len(x), None if path is None else len(path),
                                                        float(cost.max()), dt), flush=True)


def main06(only=""):
    """rrt_06 driver scene (:2012-2083): curvature 2.0, step_size 0.1, robot_radius 0.6, goal (10, 9, 0)."""
    m06 = ref_loader.load("rrt_06")
    obst = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)]      # rrt_06:2012-2014
    drv = dict(obstacles=obst, start=[0.0, 0.0, float(np.deg2rad(0.0))], goal=[10.0, 9.0, float(np.deg2rad(0.0))],
               rand_area=[-2, 15])
    for seed, it, um in ((42, 200, True), (1, 300, True), (7, 750, True), (3, 400, False), (11, 400, False)):
        n = "rrt06_drv_s%d_it%d%s" % (seed, it, "" if um else "_early")
        if n.startswith(only):
            run_rrt06(m06, n, max_iter=it, seed=seed, until_max=um, **drv)
    for tag, upd in (("noobst", dict(obstacles=[])), ("iter0", dict(max_iter=0)), ("iter1", dict(max_iter=1)),
                     ("startblocked", dict(start=[5.0, 5.0, 0.0])), ("goalinside", dict(goal=[3.0, 8.0, 0.0]))):
        n = "rrt06_edge_%s" % tag
        if n.startswith(only):
            run_rrt06(m06, n, **dict(dict(drv, max_iter=150, seed=3, until_max=True), **upd))


def main06_kat(only=""):
    """Reeds-Shepp primitive known-answer vectors: reeds_shepp_path_planning (rrt_06:1426-1441) on random pose pairs,
    the driver's curvature / step (rrt_06: curvature 1.0, step_size 0.2) and two others.  Groundwork for rrt_06
    (SURVEY 8f rank 2): pins the oracle's restatement of the solver before a planner is built on it."""
    m06 = ref_loader.load("rrt_06")
    rng = random.Random(6)
    rows = []
    for k in range(600):
        sx, sy, syaw = rng.uniform(-2, 15), rng.uniform(-2, 15), rng.uniform(-math_pi(), math_pi())
        gx, gy, gyaw = rng.uniform(-2, 15), rng.uniform(-2, 15), rng.uniform(-math_pi(), math_pi())
        if k % 7 == 0:
            gx, gy = sx + rng.uniform(-0.6, 0.6), sy + rng.uniform(-0.6, 0.6)
        if k % 11 == 0:
            gyaw = syaw
        if k % 13 == 0:
            syaw = [0.0, math_pi() / 2, -math_pi() / 2, math_pi() / 4][k % 4]
            gyaw = syaw if k % 2 else -syaw
            gx, gy = sx + (1 + k % 5), sy
        maxc = [1.0, 1.0, 0.5, 2.0][k % 4]
        step = [0.2, 0.2, 0.2, 0.1][k % 4]
        err = ""
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                px, py, pyaw, mode, lengths = m06.reeds_shepp_path_planning(sx, sy, syaw, gx, gy, gyaw, maxc, step)
            except (ZeroDivisionError, ValueError) as e:   # the reference raises on degenerate poses (u1 == 0, |arg| > 1)
                err = type(e).__name__
                px = None
        if err:
            rows.append(dict(inp=[sx, sy, syaw, gx, gy, gyaw, maxc, step], n=-1, mode=err, lengths=[], px=np.zeros(0),
                             py=np.zeros(0), pyaw=np.zeros(0)))
        elif px is None:
            rows.append(dict(inp=[sx, sy, syaw, gx, gy, gyaw, maxc, step], n=0, mode="", lengths=[], px=np.zeros(0),
                             py=np.zeros(0), pyaw=np.zeros(0)))
        else:
            rows.append(dict(inp=[sx, sy, syaw, gx, gy, gyaw, maxc, step], n=len(px), mode="".join(mode),
                             lengths=[float(v) for v in lengths], px=np.asarray(px, dtype=np.float64),
                             py=np.asarray(py, dtype=np.float64), pyaw=np.asarray(pyaw, dtype=np.float64)))
    ml = max(len(r["lengths"]) for r in rows)
    np.savez_compressed(os.path.join(OUT, "rs_kat.npz"),
                        inp=np.array([r["inp"] for r in rows]), n=np.array([r["n"] for r in rows], dtype=np.int32),
                        mode=np.array([r["mode"] for r in rows]),
                        n_len=np.array([len(r["lengths"]) for r in rows], dtype=np.int32),
                        lengths=np.array([r["lengths"] + [0.0] * (ml - len(r["lengths"])) for r in rows]),
                        poly_x=np.concatenate([r["px"] for r in rows]), poly_y=np.concatenate([r["py"] for r in rows]),
                        poly_yaw=np.concatenate([r["pyaw"] for r in rows]))
    print("rs_kat: %d cases, %d with a path, %d raise, modes %s" % (len(rows), sum(1 for r in rows if r["n"] > 0),
                                                          sum(1 for r in rows if r["n"] < 0),
                                                          sorted({r["mode"] for r in rows})[:8]), flush=True)


def math_pi():
    import math
    return math.pi


# ------------------------------------------------------------------------------------------------ rrt_08
def run_rrt08(mod, name, obstacles, start, goal, rand_area, max_iter, seed):
    """BIT* (rrt_08:138-566), driver-style BITStar(...).plan(animation=False)."""
    random.seed(seed)
    b = mod.BITStar(start=start, goal=goal, obstacleList=obstacles, randArea=rand_area, maxIter=max_iter,
                    lowerLimit=[0.0, 0.0], upperLimit=[0.0, 10.0], resolution=1.0, eta=2.0)
    tr = {"e0": [], "e1": []}
    o_best = b.best_in_edge_queue
    cm = {}
    o_is = b.informed_sample

    def best_hook():
        e = o_best()
        tr["e0"].append(float(e[0])); tr["e1"].append(float(e[1]))
        return e

    def is_hook(m, cMax, cMin, xCenter, C):
        cm["C"] = np.array(C, dtype=np.float64); cm["cMin"] = float(cMin)
        return o_is(m, cMax, cMin, xCenter, C)
    b.best_in_edge_queue = best_hook
    b.informed_sample = is_hook
    t0 = time.time()
    err = ""
    path = None
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            path = b.plan(animation=False)
        except Exception as e:  # noqa: BLE001  (the reference can raise IndexError on an empty queue)
            err = type(e).__name__
    dt = time.time() - t0
    state = random.getstate()
    vids = [float(v) for v in b.tree.vertices.keys()]
    out = dict(algo="bitstar", seed=seed, obstacles=np.array(obstacles, dtype=np.float64),
               start=np.array(start, dtype=np.float64), goal=np.array(goal, dtype=np.float64),
               rand_area=np.array(rand_area, dtype=np.float64), max_iter=max_iter,
               rot_c=cm["C"], c_min=cm["cMin"], error=err,
               path=np.array(path if path else [], dtype=np.float64).reshape(-1, 2),
               vertex_ids=np.array(vids), g_scores=np.array([float(b.g_scores[v]) for v in b.tree.vertices.keys()]),
               parent_ids=np.array([float(b.nodes.get(v, -1)) for v in b.tree.vertices.keys()]),
               n_edges=len(b.tree.edges), n_samples=len(b.samples),
               sample_ids=np.array([float(k) for k in b.samples.keys()]),
               tr_e0=np.array(tr["e0"]), tr_e1=np.array(tr["e1"]), ref_seconds=dt,
               rng_pos_after=state[1][624], rng_word0_after=np.uint32(state[1][0]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("%-24s verts=%d edges=%d path=%d samples=%d err=%r  %.2fs" % (name, len(vids), len(b.tree.edges),
          len(out["path"]), len(b.samples), err, dt), flush=True)


def main08(only=""):
    m08 = ref_loader.load("rrt_08")
    obst = [(5, 5, 0.5), (9, 6, 1), (7, 5, 1), (1, 5, 1), (3, 6, 1), (7, 9, 1)]          # rrt_08:633-635
    for seed, it, st, gl in ((42, 80, [-1.0, 0.0], [3.0, 8.0]), (1, 80, [-1.0, 0.0], [3.0, 8.0]),
                             (2, 80, [-1.0, 0.0], [3.0, 8.0]), (3, 60, [0.0, 0.0], [6.0, 10.0]),
                             (4, 40, [-1.0, 0.0], [3.0, 8.0]), (5, 80, [1.0, 1.0], [12.0, 12.0])):
        n = "rrt08_s%d_it%d" % (seed, it)
        if n.startswith(only):
            run_rrt08(m08, n, obst, st, gl, [-2, 15], it, seed)
    for tag, ob, it, st, gl in (("noobst", [], 60, [-1.0, 0.0], [3.0, 8.0]), ("iter1", obst, 1, [-1.0, 0.0], [3.0, 8.0]),
                                ("iter0", obst, 0, [-1.0, 0.0], [3.0, 8.0]),
                                # a start inside a circle is NOT here: plan() never returns (its `continue`s skip the
                                # iteration counter, :283); tests/test_gpu_parity.py checks the device ends it as OVERFLOW
                                ("goalinside", obst, 40, [-1.0, 0.0], [7.0, 9.0])):
        n = "rrt08_edge_%s" % tag
        if n.startswith(only):
            run_rrt08(m08, n, ob, st, gl, [-2, 15], it, 3)
