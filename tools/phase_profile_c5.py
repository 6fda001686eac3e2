#!/usr/bin/env python3
"""Diagnostic (GPU box): per-phase cycle shares of the RRT*-Dubins kernel from the -DRRTX_PHASE_TIMERS build.
Usage: RRTX_LIB=robotics-path-planning_amd/librrtx_prof.so python tools/phase_profile_c5.py [instances] [max_iter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import util  # noqa: E402
import rrt_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
it = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
import numpy as np  # noqa: E402

obst = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2)]
A = rrt_amd._abi
h = A.Handle(A.ALGO_DUBINS, [0.0, 0.0, 0.0], [10.0, 10.0, 0.0], [-2, 15], 3.0, 0.5, 10, it,
             robot_radius=0.0, connect_circle_dist=50.0, search_until_max_iter=True, n_instances=B,
             curvature=1.0, goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5)
h.set_obstacles(obst)
h.seed_instances(list(range(1, B + 1)))
h.plan()
s = h.get_stats()
ph = h.get_phase_cycles()
names = {0: "sample", 1: "nearest scan", 2: "ext edge (cooperative)", 3: "choose: costs + ranking", 4: "near scan", 5: "exact+dedup",
         6: "choose: lane edges + argmin", 7: "winner edge + append", 8: "rewire: lane edges", 9: "rewire: sequential",
         15: "loop", 10: "  (candidates: prepare, both stages)", 11: "  (candidates: prefix sum)",
         12: "  (candidates: points)", 13: "  (ext edge: solver on 8 lanes)", 14: "  (ext edge: points + collision)"}
tot = float(ph.sum() - ph[10] - ph[11] - ph[12] - ph[13] - ph[14])
print("instances", B, "max_iter", it, "kernel_ms", s["kernel_ms"])
for k in sorted(names):
    print("  %-30s %6.2f%%  %.1f cycles/iter/inst" % (names[k], 100.0 * ph[k] / tot if tot else 0, ph[k] / max(s["iterations"], 1)))
print("  total cycles/iter/inst %.1f" % (tot / max(s["iterations"], 1)))
print({k: s[k] for k in ("iterations", "edges_unique", "near_unique", "near_hits", "rewires", "propagated")})
