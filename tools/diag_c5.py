#!/usr/bin/env python3
"""Diagnostic (GPU box): first iteration where the rrt_05 GPU trace and the oracle trace differ for one seed."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import util  # noqa: E402
import oracle  # noqa: E402
sd = int(sys.argv[1]); it = int(sys.argv[2])
g = util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it150.npz")
g["max_iter"] = it
out = util.run_gpu_dubins(g, [sd], trace_instance=0)
r = oracle.plan_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], it, seed=sd, trace=True)
rx, ry, ne, nn = out["trace"]
n = min(len(ne), len(r["tr_nearest"]))
d = np.nonzero((ne[:n] != r["tr_nearest"][:n]) | (nn[:n] != r["tr_n_near"][:n]) | (rx[:n] != r["tr_rx"][:n]))[0]
print("iterations", n, "first differing iteration", d[:1])
if len(d):
    i = d[0]
    print("gpu   nearest", ne[i], "n_near", nn[i], "rx", rx[i])
    print("orcl  nearest", r["tr_nearest"][i], "n_near", r["tr_n_near"][i], "rx", r["tr_rx"][i])
    print("prev  n_near gpu/orcl", nn[i-3:i], r["tr_n_near"][i-3:i])
print("stats gpu", {k: out["stats"][k] for k in ("edges_unique", "near_hits", "near_unique", "rewires", "propagated")})
print("stats orc", {k: r["stats"][k] for k in ("edges_unique", "near_hits", "near_unique", "rewires", "propagated")})
