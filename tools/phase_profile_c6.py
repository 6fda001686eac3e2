#!/usr/bin/env python3
"""Diagnostic (GPU box): cycle shares of the stages of the cooperative Reeds-Shepp steer in the rrt_06 kernel, from the
-DRRTX_PHASE_TIMERS build.  Usage: RRTX_LIB=robotics-path-planning_amd/librrtx_prof.so python tools/phase_profile_c6.py [instances] [max_iter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import rrt_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
it = int(sys.argv[2]) if len(sys.argv) > 2 else 750
obst = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)]
A = rrt_amd._abi
h = A.Handle(A.ALGO_RS, [0.0, 0.0, 0.0], [10.0, 9.0, 0.0], [-2, 15], 3.0, 0.5, 10, it, robot_radius=0.6,
             connect_circle_dist=50.0, search_until_max_iter=True, n_instances=B, curvature=2.0,
             goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5, step_size=0.1)
h.set_obstacles(obst)
h.seed_instances(list(range(1, B + 1)))
h.plan()
s = h.get_stats()
ph = h.get_phase_cycles()
names = {0: "steer: 48 word variants", 1: "steer: set_path + arg-min", 2: "steer: course layout",
         3: "steer: points + collision + store"}
main = {4: "sample", 5: "nearest scans", 6: "extension edge", 7: "near scans (2 stages)", 8: ".index collapse",
        9: "choose_parent (costs, arg-min, edges)", 10: "rewire (search + edges)", 11: "propagate", 12: "try_goal_path edge",
        13: "loop glue / trace / early test"}
tot = float(ph[15])
print("instances", B, "max_iter", it, "kernel_ms", s["kernel_ms"])
for k in sorted(names):
    print("  %-36s %6.2f%%  %.0f ticks/edge call" % (names[k], 100.0 * ph[k] / tot if tot else 0, ph[k] / max(s["edges_unique"], 1)))
print(" main loop (the steer stages above are inside these):")
for k in sorted(main):
    print("  %-40s %6.2f%%  %.0f ticks/iteration" % (main[k], 100.0 * ph[k] / tot if tot else 0, ph[k] / max(s["iterations"], 1)))
print("  %-40s %6.2f%%" % ("outside the loop", 100.0 * (tot - ph[4:14].sum()) / tot if tot else 0))
print({k: s[k] for k in ("iterations", "edges_unique", "near_hits", "rewires", "propagated", "total_nodes")})
