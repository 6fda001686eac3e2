#!/usr/bin/env python3
"""Diagnostic (GPU box): rrt_05 batch on the GPU vs the oracle, instance by instance.
Usage: python tools/check_c5_batch.py [first_seed] [count] [max_iter]"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import util  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
it = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
g = util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it150.npz")
g["max_iter"] = it


def orc(sd):
    import oracle
    r = oracle.plan_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], it, seed=sd)
    return sd, r["x"], r["y"], r["cost"], r["parent"]


if __name__ == "__main__":
    seeds = list(range(first, first + cnt))
    try:
        out = util.run_gpu_dubins(g, seeds)
    except Exception as e:  # noqa: BLE001
        print("GPU run failed:", e)
        sys.exit(1)
    bad = 0
    with ProcessPoolExecutor(max_workers=12) as ex:
        for sd, x, y, cost, parent in ex.map(orc, seeds):
            t = out["trees"][sd - first]
            ok = np.array_equal(t[0], x) and np.array_equal(t[1], y) and np.array_equal(t[2], cost) and np.array_equal(t[3], parent)
            if not ok:
                bad += 1
                n = min(len(x), len(t[0]))
                dif = np.nonzero((t[0][:n] != x[:n]) | (t[3][:n] != parent[:n]))[0]
                print("MISMATCH seed", sd, "nodes gpu/oracle", len(t[0]), len(x), "first differing node", dif[:1])
    print("checked", cnt, "instances x", it, "iterations: mismatches", bad)
