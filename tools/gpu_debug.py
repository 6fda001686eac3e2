#!/usr/bin/env python3
"""Debug harness (GPU box): run golden scenarios through the HIP path with the per-iteration trace on and
print where it first leaves the reference's trajectory."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import util  # noqa: E402

names = sys.argv[1:] or ["rrt04_drv_mt_s1234", "rrt01_drv_s42", "rrt04_drv_sobol_s1234", "rrt04_drv_early_s1",
                         "rrt04_c2_s1_it1000", "rrt04_c2_s1_it4000"]
for nm in names:
    g = util.load_golden(os.path.join(util.GOLDEN, nm + ".npz"))
    kw = util.kwargs_from_golden(g)
    t0 = time.time()
    try:
        out = util.run_gpu_batch(kw, [int(g["seed"])], trace_instance=0)
    except Exception as e:  # noqa: BLE001
        print(nm, "FAILED:", e)
        continue
    dt = time.time() - t0
    x, y, cost, parent = out["trees"][0]
    d = util.first_trace_divergence(out["trace"], g["tr_rnd_x"], g["tr_rnd_y"], g["tr_nearest"])
    same = len(x) == len(g["x"]) and np.array_equal(parent, g["parent"]) and np.array_equal(x, g["x"]) \
        and np.array_equal(y, g["y"]) and (kw["algo"] == "rrt" or np.array_equal(cost, g["cost"]))
    p = out["paths"][0]
    psame = (p is None and len(g["path"]) == 0) or (p is not None and np.array_equal(p, g["path"]))
    print("%-26s n=%d/%d tree_equal=%s path_equal=%s first_trace_div=%s edges_ref=%d/%d  %.2fs kernel=%.1fms launches=%d"
          % (nm, len(x), len(g["x"]), same, psame, d, out["stats"]["edges_ref"], int(g["ref_edges"]), dt,
             out["stats"]["kernel_ms"], out["stats"]["launches"]))
    if d is not None:
        tr = out["trace"]
        for i in range(max(0, d - 1), min(len(tr[0]), d + 2)):
            print("   it %d gpu rnd=(%r,%r) nearest=%d nn=%d | ref rnd=(%r,%r) nearest=%d"
                  % (i, tr[0][i], tr[1][i], tr[2][i], tr[3][i], g["tr_rnd_x"][i], g["tr_rnd_y"][i], g["tr_nearest"][i]))
    elif not same:
        m = min(len(x), len(g["x"]))
        bad = np.nonzero((parent[:m] != g["parent"][:m]) | (x[:m] != g["x"][:m]) | (y[:m] != g["y"][:m]) |
                         (cost[:m] != g["cost"][:m]))[0]
        print("   first differing node", bad[:5], "of", m)
        for b in bad[:3]:
            print("   node %d gpu (%r,%r,c=%r,p=%d) ref (%r,%r,c=%r,p=%d)" % (b, x[b], y[b], cost[b], parent[b],
                  g["x"][b], g["y"][b], g["cost"][b], g["parent"][b]))
        # n_near trace
        nn_ref = g["tr_n_near"]
        acc = [k for k in out["trace"][3] if k >= 0]
        print("   sum n_near_unique gpu", sum(acc), "ref hits", int(nn_ref.sum()))
