#!/usr/bin/env python3
"""Golden vectors for Node.path_x / path_y (the draw data of rrt_01 / rrt_04 nodes, rrt_04:1165-1167): RUNS the reference
planners (build container only, through oracle/ref_loader.py) on their drivers' scenes and stores, per node, the polyline
the reference holds after planning().  Data only; written to tests/golden/nodepaths_*.npz.
Usage: python oracle/gen_golden_paths.py"""
import contextlib
import io
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_loader  # noqa: E402
from gen_golden import DRIVER_OBST, OUT, tree_arrays  # noqa: E402


def run(short, name, seed, **kw):
    mod = ref_loader.load(short)
    ref_loader.reset_sobol(mod)
    random.seed(seed)
    rrt = mod.RRT(**kw)
    with contextlib.redirect_stdout(io.StringIO()):
        path = rrt.planning(animation=False)
    nl = rrt.node_list
    x, y, cost, parent = tree_arrays(nl)
    plen = np.array([len(nd.path_x) for nd in nl], dtype=np.int32)
    ppx = np.concatenate([np.asarray(nd.path_x, dtype=np.float64).reshape(-1) for nd in nl])
    ppy = np.concatenate([np.asarray(nd.path_y, dtype=np.float64).reshape(-1) for nd in nl])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, x=x, y=y, parent=parent, path_len=plen, path_x=ppx,
                        path_y=ppy, path=np.zeros((0, 2)) if path is None else np.array(path, dtype=np.float64),
                        **{"kw_" + k: np.array(v if v is not None else [], dtype=np.float64) for k, v in kw.items()
                           if k != "obstacle_list"}, obstacles=np.array(kw["obstacle_list"], dtype=np.float64))
    print(name, len(nl), "nodes", int(plen.sum()), "points", "rewired", int((parent > np.arange(len(nl))).sum()))


if __name__ == "__main__":
    common = dict(start=[0, 0], goal=[6.0, 10.0], obstacle_list=DRIVER_OBST, rand_area=[-2, 15], expand_dis=1.0,
                  path_resolution=0.1, goal_sample_rate=5, max_iter=500, play_area=[0, 10, 0, 14], robot_radius=0.6)
    run("rrt_01", "nodepaths_rrt01_s42", 42, **common)
    for sd, until in ((1234, True), (7, True), (3, False)):
        run("rrt_04", "nodepaths_rrt04_s%d_%s" % (sd, "full" if until else "early"), sd, sobol_sampler=False,
            connect_circle_dist=50.0, search_until_max_iter=until, **common)
    run("rrt_04", "nodepaths_rrt04_res03_s5", 5, sobol_sampler=False, connect_circle_dist=50.0, search_until_max_iter=True,
        **dict(common, expand_dis=3.0, path_resolution=0.3, play_area=None, robot_radius=0.0, max_iter=300))
