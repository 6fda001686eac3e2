#!/usr/bin/env python3
"""Full-size parity of the EXACT bench workload (GPU box): the C2 batch as bench.py plans it -- 4 096 instances x 105 000
iterations in one handle (64-thread shape, 16-bit first-stage mirror, one launch) -- with a sample of its trees compared
against the golden-pinned oracle bit for bit (x, y, cost, parent, path).  The suite holds the same comparison for one
full-size instance on the same kernel shape and for 32 instances of a 3 072 x 3 000 batch; this is the one-off check
of the bench's own batch at its own size (the oracle needs ~100 s per 105 000-iteration tree on one core).
Usage: python tools/full_size_parity.py [n_sampled=16] [instances=4096] [max_iter=105000]   -> profiles/r2_full_size_parity.txt"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import util  # noqa: E402

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
MI = int(sys.argv[3]) if len(sys.argv) > 3 else 105000
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
