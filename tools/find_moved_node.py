#!/usr/bin/env python3
"""CPU search (oracle) for planning problems that reach rrt_04 rewire's rare branch: a node MOVED by an unsnapped
steer (rrt_04:1372 with :1105-1110) and, rarer still, listed again in near_inds through a distance tie (:1337) so that
the reference visits it a second time.  Prints (config, seed, moved, revisit) for every hit.
Usage: python tools/find_moved_node.py [n_seeds] [iterations]"""
import concurrent.futures as cf
import ctypes as C
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def job(a):
    import oracle
    import util
    (res, exp, rate, ccd, scene), seed, iters = a
    L = oracle.lib()
    m0, r0 = C.c_long(), C.c_long()
    L.orc_moved_counters(C.byref(m0), C.byref(r0))
    kw = dict(util.C2)
    if scene == "line":
        kw.update(start=[0, 0], goal=[10, 0], rand_area=[-2, 12], obstacles=[])
    elif scene == "diag":
        kw.update(start=[0, 0], goal=[6, 8], rand_area=[-2, 12], obstacles=[(3, 3, 1)])
    else:
        kw.update(start=[0, 0], goal=[6, 10], rand_area=[-2, 15],
                  obstacles=[(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)])
    kw.update(expand_dis=exp, path_resolution=res, goal_sample_rate=rate, connect_circle_dist=ccd, max_iter=iters,
              robot_radius=0.0)
    oracle.plan(seed=seed, exact_pow=True, **kw)
    m1, r1 = C.c_long(), C.c_long()
    L.orc_moved_counters(C.byref(m1), C.byref(r1))
    return (res, exp, rate, ccd, scene), seed, m1.value - m0.value, r1.value - r0.value


def main():
    nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    cfgs = list(itertools.product([0.1, 0.3, 0.7, 0.05], [1.0, 2.1, 3.0], [20, 60, 95], [50.0], ["line", "diag", "drv"]))
    jobs = [(c, s, iters) for c in cfgs for s in range(1, nseed + 1)]
    tot_m = tot_r = 0
    with cf.ProcessPoolExecutor(max_workers=6) as ex:
        for cfg, seed, m, r in ex.map(job, jobs, chunksize=8):
            tot_m += m
            tot_r += r
            if m:
                print("cfg", cfg, "seed", seed, "moved", m, "revisit", r, flush=True)
    print("total moved", tot_m, "revisit", tot_r, "over", len(jobs), "plans")


if __name__ == "__main__":
    main()
