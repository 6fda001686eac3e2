#!/usr/bin/env python3
"""Full-size parity of the EXACT bench workload (GPU box): the C2 batch as bench.py plans it -- 4 096 instances x 105 000
iterations in one handle (64-thread shape, 16-bit first-stage mirror, one launch) -- with a sample of its trees compared
against the golden-pinned oracle bit for bit (x, y, cost, parent, path).  The suite holds the same comparison for one
full-size instance on the same kernel shape and for 32 instances of a 3 072 x 3 000 batch; this is the one-off check
of the bench's own batch at its own size (the oracle needs ~100 s per 105 000-iteration tree on one core).
Usage: python tools/full_size_parity.py [n_sampled=16] [instances=4096] [max_iter=105000]   -> profiles/r2_full_size_parity.txt
       python tools/full_size_parity.py c3|c5 [n_sampled=64]    the C3 (rrt_07, 1 024 x 20 000) / C5 (rrt_05, 1 536 x 5 000) bench
       batches, built by bench.py's own Workload class, sampled trees against the oracle"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import util  # noqa: E402

def orc_other(a):
    import oracle
    w, kw, it, sd = a
    if w == "c3":
        kc = dict(kw)
        kc.pop("algo")
        kc["max_iter"] = it
        r = oracle.plan_informed(seed=sd, exact_pow=False, **kc)
        return r["x"], r["y"], r["cost"], r["parent"], r["path"], None
    r = oracle.plan_dubins(kw["start"], kw["goal"], kw["obstacles"], kw["rand_area"], it, seed=sd)
    return r["x"], r["y"], r["cost"], r["parent"], r["path"], r["yaw"]


def other_workload(w, ns):
    """C3 / C5 exactly as bench.py builds them."""
    import importlib
    import types
    import bench
    import rrt_amd
    sharding = importlib.import_module("robotics-path-planning_amd.sharding")
    a = types.SimpleNamespace(workload=w, instances=None, max_iter=None, obstacles=None, warmup_max_iter=0)
    wl = bench.Workload(a, np, util, rrt_amd)
    wl.prepare(0, sharding)
    pick = sorted(set(int(v) for v in np.linspace(0, wl.B - 1, ns)))
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(16, len(pick))) as ex:
        fut = [ex.submit(orc_other, (w, wl.kw, wl.max_iter, wl.seeds[i])) for i in pick]
        h = wl.make_handle(wl.max_iter, 0)
        tg = time.time()
        wl.step(h)
        tg = time.time() - tg
        st = h.get_stats()
        print("GPU %s: %d instances x %d iterations planned in %.2f s (kernel %.2f s; edges steered / tested by the device %d, "
              "reference-equivalent %d)" % (w, wl.B, wl.max_iter, tg, st["kernel_ms"] / 1e3, st["edges_unique"], st["edges_ref"]),
              flush=True)
        pc, nn, status = h.get_results()
        bad = 0
        for i, f in zip(pick, fut):
            ox, oy, oc, op, opath, oyaw = f.result()
            x, y, c, p = h.get_tree(i)
            path = h.get_path(i)
            ok = (len(x) == len(ox) and np.array_equal(x, ox) and np.array_equal(y, oy) and np.array_equal(c, oc)
                  and np.array_equal(p, op) and ((path is None) == (opath is None))
                  and (path is None or np.array_equal(np.asarray(path)[:, :2], np.asarray(opath)[:, :2])))
            if ok and oyaw is not None:
                ok = np.array_equal(h.get_yaw(i), oyaw)
            bad += 0 if ok else 1
            print("instance %4d (seed %4d): %6d nodes, path cost %.12f -> %s"
                  % (i, wl.seeds[i], len(x), pc[i], "identical" if ok else "MISMATCH"), flush=True)
        h.close()
    print("%s: sampled %d of %d trees: mismatches %d (wall %.0f s)" % (w, len(pick), wl.B, bad, time.time() - t0))
    return bad


if len(sys.argv) > 1 and sys.argv[1] in ("c3", "c5"):
    if __name__ == "__main__":
        sys.exit(1 if other_workload(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64) else 0)
    NS = 0
else:
    NS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 and NS else 4096
MI = int(sys.argv[3]) if len(sys.argv) > 3 and NS else 105000
KW = util.c2_kwargs(MI)


def orc(sd):
    r = util.run_oracle(KW, sd, exact_pow=False)
    return r["x"], r["y"], r["cost"], r["parent"], r["path"], r["stats"]["edges_ref"]


if __name__ == "__main__":
    import rrt_amd
    A = rrt_amd._abi
    seeds = list(range(1, B + 1))
    pick = sorted(set(int(v) for v in np.linspace(0, B - 1, NS)))
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(16, len(pick))) as ex:   # the oracle runs while the GPU plans
        fut = [ex.submit(orc, seeds[i]) for i in pick]
        h = A.Handle(A.ALGO_RRT_STAR, KW["start"], KW["goal"], KW["rand_area"], KW["expand_dis"], KW["path_resolution"],
                     KW["goal_sample_rate"], KW["max_iter"], robot_radius=0.0, connect_circle_dist=50.0,
                     search_until_max_iter=True, n_instances=B)
        h.set_obstacles(KW["obstacles"])
        h.seed_instances(seeds)
        tg = time.time()
        rc = h.plan()
        tg = time.time() - tg
        st = h.get_stats()
        print("GPU: %d instances x %d iterations planned in %.1f s (rc %d; kernel %.1f s; f32 / q16 fallbacks %d / %d; replanned %d)"
              % (B, MI, tg, rc, st["kernel_ms"] / 1e3, st.get("f32_fallbacks", 0), st.get("q16_fallbacks", 0),
                 st.get("replanned", 0)), flush=True)
        pc, nn, status = h.get_results()
        bad = 0
        for i, f in zip(pick, fut):
            ox, oy, oc, op, opath, oe = f.result()
            x, y, c, p = h.get_tree(i)
            path = h.get_path(i)
            ok = (len(x) == len(ox) and np.array_equal(x, ox) and np.array_equal(y, oy) and np.array_equal(c, oc)
                  and np.array_equal(p, op) and ((path is None) == (opath is None))
                  and (path is None or np.array_equal(path, opath)))
            bad += 0 if ok else 1
            print("instance %4d (seed %4d): %6d nodes, path cost %.12f, oracle nodes %6d -> %s"
                  % (i, seeds[i], len(x), pc[i], len(ox), "identical" if ok else "MISMATCH"), flush=True)
        h.close()
    print("sampled %d of %d trees: mismatches %d (wall %.0f s)" % (len(pick), B, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
