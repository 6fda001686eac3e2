#!/usr/bin/env python3
"""Diagnostic (GPU box): per-phase cycle shares of the Informed RRT* kernel from the -DRRTX_PHASE_TIMERS build.
Usage: RRTX_LIB=robotics-path-planning_amd/librrtx_prof.so python tools/phase_profile_c3.py [instances] [max_iter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import util  # noqa: E402
import rrt_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
it = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
A = rrt_amd._abi
start, goal = [2, 2], [98, 98]
c_min, c = rrt_amd.informed_rotation(start, goal)
h = A.Handle(A.ALGO_INFORMED, start, goal, [0, 100], 0.5, 1.0, 10, it, sampler=A.SAMPLER_SOBOL, n_instances=B,
             informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
h.set_obstacles(util.synth_map(11, 200, 0.3, 1.5))
h.seed_instances(list(range(1, B + 1)))
h.plan()
s = h.get_stats()
ph = h.get_phase_cycles()
names = {0: "sample (when not read ahead)", 1: "nearest (prefetched answer / own pass / 2nd pass)", 2: "steer + extension collision (+ read-ahead draw)",
         3: "fused 16-bit pass (near + next nearest)", 4: "candidates: exact re-check + de-dup", 5: "choose_parent: hypot + rank",
         6: "choose_parent: collision batches", 9: "append + rewire", 15: "goal bookkeeping + commit + loop",
         7: "choose_parent: hypot + cost sums", 10: "steer: x[ni] + atan2 + cos | sin | read-ahead", 11: "steer: node, hypot, goal test",
         12: "pass set-up (radius, grid point)"}
tot = float(ph.sum())
print("instances", B, "max_iter", it, "kernel_ms", s["kernel_ms"])
for k in sorted(names):
    print("  %-30s %6.2f%%  %.1f cycles/iter/inst" % (names[k], 100.0 * ph[k] / tot if tot else 0, ph[k] / max(s["iterations"], 1)))
print("  total cycles/iter/inst %.1f" % (tot / max(s["iterations"], 1)))
print({k: s[k] for k in ("iterations", "edges_unique", "near_unique", "near_hits", "rewires", "q16_fallbacks", "exact_rescans")})
