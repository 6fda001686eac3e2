"""Shared helpers of the test-suite: golden loading, oracle/GPU invocation, comparisons."""
import glob
import os
import random

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_files(prefix=""):
    """rrt_01 / rrt_04 goldens by default; prefix="rrt07" selects the Informed RRT* ones."""
    fs = sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))
    if not prefix:
        fs = [f for f in fs if os.path.basename(f).startswith(("rrt01", "rrt02", "rrt04"))]
    return fs


def load_golden(path):
    g = np.load(path)
    d = {k: g[k] for k in g.files}
    d["name"] = os.path.basename(path)[:-4]
    return d


def kwargs_from_golden(g):
    return dict(algo=str(g["algo"]), start=[float(v) for v in g["start"]], goal=[float(v) for v in g["goal"]],
                obstacles=[tuple(float(v) for v in o) for o in g["obstacles"]],
                rand_area=[float(v) for v in g["rand_area"]], expand_dis=float(g["expand_dis"]),
                path_resolution=float(g["path_resolution"]), goal_sample_rate=int(g["goal_sample_rate"]),
                max_iter=int(g["max_iter"]), play_area=[float(v) for v in g["play_area"]] if len(g["play_area"]) else None,
                robot_radius=float(g["robot_radius"]), sobol=int(g["sobol"]),
                connect_circle_dist=float(g["connect_circle_dist"]), search_until_max_iter=int(g["until_max"]))


def run_oracle(kw, seed, exact_pow=True, trace=False):
    import oracle
    return oracle.plan(seed=seed, exact_pow=exact_pow, trace=trace, **kw)


def synth_map(map_seed, m, rmin=0.5, rmax=2.5):
    """SURVEY.md 8(d) map generator (same as oracle/gen_golden.py)."""
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


C2 = dict(algo="rrt_star", start=[2, 2], goal=[98, 98], rand_area=[0, 100], expand_dis=2.0, path_resolution=0.25,
          goal_sample_rate=5, play_area=None, robot_radius=0.0, sobol=0, connect_circle_dist=50.0,
          search_until_max_iter=1)


def c2_kwargs(max_iter, m=50, map_seed=7):
    kw = dict(C2)
    kw["obstacles"] = synth_map(map_seed, m)
    kw["max_iter"] = max_iter
    return kw


def run_gpu_batch(kw, seeds, device=0, trace_instance=None):
    """Plan len(seeds) instances on the GPU through the C ABI; returns (handle-free) result dict."""
    import rrt_amd
    A = rrt_amd._abi
    h = A.Handle({"rrt": A.ALGO_RRT, "rrt_star": A.ALGO_RRT_STAR}[kw["algo"]], kw["start"], kw["goal"], kw["rand_area"],
                 kw["expand_dis"], kw["path_resolution"], kw["goal_sample_rate"], kw["max_iter"],
                 play_area=kw["play_area"], robot_radius=kw["robot_radius"],
                 sampler=A.SAMPLER_SOBOL if kw["sobol"] else A.SAMPLER_MT,
                 connect_circle_dist=kw["connect_circle_dist"], search_until_max_iter=kw["search_until_max_iter"],
                 n_instances=len(seeds), device=device)
    try:
        h.set_obstacles(kw["obstacles"])
        h.seed_instances(seeds)
        if trace_instance is not None:
            h.enable_trace(trace_instance)
        h.plan(strict=True)
        out = dict(stats=h.get_stats(), results=h.get_results(), trees=[], paths=[], rng=[])
        for i in range(len(seeds)):
            out["trees"].append(h.get_tree(i))
            out["paths"].append(h.get_path(i))
            out["rng"].append(h.get_rng_state(i))
        if trace_instance is not None:
            out["trace"] = h.get_trace()
        if kw["sobol"]:
            out["sobol_index"] = [h.get_sobol_index(i) for i in range(len(seeds))]
    finally:
        h.close()
    return out


def first_trace_divergence(tr_gpu, rx, ry, nearest):
    n = min(len(tr_gpu[0]), len(rx))
    bad = np.nonzero((tr_gpu[0][:n] != rx[:n]) | (tr_gpu[1][:n] != ry[:n]) | (tr_gpu[2][:n] != nearest[:n]))[0]
    return int(bad[0]) if len(bad) else None


def assert_tree_equal(got, want, what=""):
    """got/want: (x, y, cost, parent); ints exact, doubles bit-exact (stronger than the 1e-6 contract)."""
    gx, gy, gc, gp = got
    wx, wy, wc, wp = want
    assert len(gx) == len(wx), "%s: node count %d != %d" % (what, len(gx), len(wx))
    assert np.array_equal(gp, wp), "%s: parent[] differs first at %d" % (what, int(np.nonzero(gp != wp)[0][0]))
    assert np.array_equal(gx, wx) and np.array_equal(gy, wy), "%s: coordinates differ" % what
    if wc is not None:
        assert np.array_equal(gc, wc), "%s: cost[] differs (max abs %g)" % (what, float(np.abs(gc - wc).max()))


def informed_kwargs_from_golden(g):
    return dict(start=[float(v) for v in g["start"]], goal=[float(v) for v in g["goal"]],
                obstacles=[tuple(float(v) for v in o) for o in g["obstacles"]],
                rand_area=[float(v) for v in g["rand_area"]], expand_dis=float(g["expand_dis"]),
                goal_sample_rate=int(g["goal_sample_rate"]), max_iter=int(g["max_iter"]), sobol=int(g["sobol"]))


def run_gpu_informed(kw, seeds, device=0, trace_instance=None):
    """Informed RRT* (rrt_07) instances on the GPU through the C ABI."""
    import rrt_amd
    A = rrt_amd._abi
    c_min, c = rrt_amd.informed_rotation(kw["start"], kw["goal"])
    h = A.Handle(A.ALGO_INFORMED, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], 1.0,
                 kw["goal_sample_rate"], kw["max_iter"], sampler=A.SAMPLER_SOBOL if kw["sobol"] else A.SAMPLER_MT,
                 n_instances=len(seeds), device=device, informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]],
                 informed_c_min=c_min)
    try:
        h.set_obstacles(kw["obstacles"])
        h.seed_instances(seeds)
        if trace_instance is not None:
            h.enable_trace(trace_instance)
        h.plan(strict=True)
        out = dict(stats=h.get_stats(), results=h.get_results(), trees=[], paths=[], rng=[])
        for i in range(len(seeds)):
            out["trees"].append(h.get_tree(i))
            out["paths"].append(h.get_path(i))
            out["rng"].append(h.get_rng_state(i))
        if trace_instance is not None:
            out["trace"] = h.get_trace()
        if kw["sobol"]:
            out["sobol_index"] = [h.get_sobol_index(i) for i in range(len(seeds))]
    finally:
        h.close()
    return out


def run_gpu_dubins(g, seeds, device=0, trace_instance=None, max_iter=None):
    """RRT*-Dubins (rrt_05) instances on the GPU through the C ABI; g = golden dict (parameters)."""
    import rrt_amd
    A = rrt_amd._abi
    mi = int(g["max_iter"]) if max_iter is None else max_iter
    h = A.Handle(A.ALGO_DUBINS, [float(v) for v in g["start"]], [float(v) for v in g["goal"]],
                 [float(v) for v in g["rand_area"]], float(g["expand_dis"]), 0.5, int(g["goal_sample_rate"]), mi,
                 robot_radius=float(g["robot_radius"]), connect_circle_dist=float(g["connect_circle_dist"]),
                 search_until_max_iter=bool(int(g.get("search_until_max_iter", 1))), n_instances=len(seeds),
                 device=device, curvature=float(g["curvature"]),
                 goal_yaw_th=float(g["goal_yaw_th"]), goal_xy_th=float(g["goal_xy_th"]))
    try:
        h.set_obstacles([tuple(float(v) for v in o) for o in g["obstacles"]])
        h.seed_instances(seeds)
        if trace_instance is not None:
            h.enable_trace(trace_instance)
        h.plan(strict=True)
        out = dict(stats=h.get_stats(), results=h.get_results(), trees=[], yaws=[], polys=[], paths=[], rng=[])
        for i in range(len(seeds)):
            out["trees"].append(h.get_tree(i))
            out["yaws"].append(h.get_yaw(i))
            out["polys"].append(h.get_polylines(i))
            out["paths"].append(h.get_path(i))
            out["rng"].append(h.get_rng_state(i))
        if trace_instance is not None:
            out["trace"] = h.get_trace()
    finally:
        h.close()
    return out


def run_gpu_rrt_dubins(g, seeds, device=0, trace_instance=None):
    """RRT with Dubins steer (rrt_03) instances on the GPU through the C ABI; g = golden dict (parameters)."""
    import rrt_amd
    A = rrt_amd._abi
    h = A.Handle(A.ALGO_RRT_DUBINS, [float(v) for v in g["start"]], [float(v) for v in g["goal"]],
                 [float(v) for v in g["rand_area"]], 0.0, 0.5, int(g["goal_sample_rate"]), int(g["max_iter"]),
                 robot_radius=float(g["robot_radius"]), sampler=A.SAMPLER_SOBOL if int(g["sobol"]) else A.SAMPLER_MT,
                 search_until_max_iter=bool(int(g.get("search_until_max_iter", 1))), n_instances=len(seeds),
                 device=device, curvature=float(g["curvature"]),
                 goal_yaw_th=float(g["goal_yaw_th"]), goal_xy_th=float(g["goal_xy_th"]))
    try:
        h.set_obstacles([tuple(float(v) for v in o) for o in g["obstacles"]])
        h.seed_instances(seeds)
        if trace_instance is not None:
            h.enable_trace(trace_instance)
        h.plan(strict=True)
        out = dict(stats=h.get_stats(), results=h.get_results(), trees=[], yaws=[], polys=[], paths=[], rng=[], sobol=[])
        for i in range(len(seeds)):
            out["trees"].append(h.get_tree(i))
            out["yaws"].append(h.get_yaw(i))
            out["polys"].append(h.get_polylines(i))
            out["paths"].append(h.get_path(i))
            out["rng"].append(h.get_rng_state(i))
            out["sobol"].append(h.get_sobol_index(i))
        if trace_instance is not None:
            out["trace"] = h.get_trace()
    finally:
        h.close()
    return out


def run_gpu_rrt_rs(g, seeds, device=0, trace_instance=None):
    """RRT*-Reeds-Shepp (rrt_06) instances on the GPU through the C ABI; g = golden dict (parameters)."""
    import rrt_amd
    A = rrt_amd._abi
    h = A.Handle(A.ALGO_RS, [float(v) for v in g["start"]], [float(v) for v in g["goal"]],
                 [float(v) for v in g["rand_area"]], float(g["expand_dis"]), 0.5, 10, int(g["max_iter"]),
                 robot_radius=float(g["robot_radius"]), connect_circle_dist=float(g["connect_circle_dist"]),
                 search_until_max_iter=bool(int(g.get("search_until_max_iter", 1))), n_instances=len(seeds),
                 device=device, curvature=float(g["curvature"]), goal_yaw_th=float(g["goal_yaw_th"]),
                 goal_xy_th=float(g["goal_xy_th"]), step_size=float(g["step_size"]))
    try:
        h.set_obstacles([tuple(float(v) for v in o) for o in g["obstacles"]])
        h.seed_instances(seeds)
        if trace_instance is not None:
            h.enable_trace(trace_instance)
        h.plan(strict=True)
        out = dict(stats=h.get_stats(), results=h.get_results(), trees=[], yaws=[], polys=[], paths=[], path_yaws=[], rng=[])
        for i in range(len(seeds)):
            out["trees"].append(h.get_tree(i))
            out["yaws"].append(h.get_yaw(i))
            out["polys"].append(h.get_polylines(i))
            out["paths"].append(h.get_path(i))
            out["path_yaws"].append(h.get_path_yaw(i))
            out["rng"].append(h.get_rng_state(i))
        if trace_instance is not None:
            out["trace"] = h.get_trace()
    finally:
        h.close()
    return out


def run_gpu_bitstar(obstacles, rand_area, max_iter, seeds, starts, goals, device=0, trace_instance=None):
    """BIT* (rrt_08) instances on the GPU through the C ABI; per-instance start / goal / rotation."""
    import rrt_amd
    A = rrt_amd._abi
    c_min, c = rrt_amd.bitstar_rotation(starts[0], goals[0])
    h = A.Handle(A.ALGO_BITSTAR, starts[0], goals[0], rand_area, 2.0, 1.0, 0, max_iter, n_instances=len(seeds),
                 device=device, informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
    try:
        h.set_obstacles(obstacles)
        h.seed_instances(seeds)
        for i in range(len(seeds)):
            cm, ci = rrt_amd.bitstar_rotation(starts[i], goals[i])
            h.set_instance(i, starts[i], goals[i])
            h.set_instance_rotation(i, [ci[0, 0], ci[0, 1], ci[1, 0], ci[1, 1]], cm)
        if trace_instance is not None:
            h.enable_trace(trace_instance)
        h.plan(strict=True)
        out = dict(stats=h.get_stats(), results=h.get_results(), trees=[], paths=[], rng=[])
        for i in range(len(seeds)):
            out["trees"].append(h.get_tree(i))
            out["paths"].append(h.get_path(i))
            out["rng"].append(h.get_rng_state(i))
        if trace_instance is not None:
            out["trace"] = h.get_trace()
    finally:
        h.close()
    return out
