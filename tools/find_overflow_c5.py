#!/usr/bin/env python3
"""Diagnostic (GPU box): which rrt_05 seeds end with the overflow status (GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import util, rrt_amd
A = rrt_amd._abi
g = util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it150.npz")
first, cnt, it = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seeds = list(range(first, first + cnt))
h = A.Handle(A.ALGO_DUBINS, [float(v) for v in g["start"]], [float(v) for v in g["goal"]], [float(v) for v in g["rand_area"]],
             3.0, 0.5, 10, it, robot_radius=0.0, connect_circle_dist=50.0, search_until_max_iter=True, n_instances=cnt,
             curvature=1.0, goal_yaw_th=float(g["goal_yaw_th"]), goal_xy_th=0.5)
h.set_obstacles([tuple(float(v) for v in o) for o in g["obstacles"]])
h.seed_instances(seeds)
try:
    h.plan()
except Exception as e:
    print("plan:", str(e)[:80])
pc, nn, st = h.get_results()
bad = [(seeds[i], int(nn[i])) for i in range(cnt) if st[i] & 4]
print("overflow instances:", bad[:10], "count", len(bad))
