#!/usr/bin/env python3
"""CPU baseline at FULL size (SURVEY 8d ii): one whole plan of a workload's own tree size on one host thread, with the
golden-pinned oracle (oracle/rrt_oracle.c) -- minutes of CPU time, so it is measured once per CPU model and kept under
profiles/cpu_fullsize_<workload>.json; bench.py's cpu_baseline reports it beside the bounded sample it times live.
Usage (GPU box, through gpurun): python3 tools/cpu_fullsize.py c2 [c3 c5 ...]   -> profiles/ + gpurun_out/"""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import util  # noqa: E402
import bench  # noqa: E402


class _NoAbi:
    """bench.Workload only needs the package for handle construction, which is not used here."""
    _abi = None


def main():
    for w in sys.argv[1:] or ["c2"]:
        a = types.SimpleNamespace(workload=w, instances=None, max_iter=None, obstacles=None, warmup_max_iter=0)
        wl = bench.Workload(a, np, util, _NoAbi)
        seed = 1
        t = time.perf_counter()
        eu, er, plans, _ = bench._cpu_job((w, wl.kw, wl.max_iter, seed, None))
        dt = time.perf_counter() - t
        rec = {"workload": w, "max_iter": wl.max_iter, "obstacles": wl.M, "seed": seed, "seconds": dt,
               "edges_unique": eu, "edges_ref": er, "value": eu / dt, "reference_equivalent_value": er / dt,
               "unit": "edge expansions/s", "threads": 1, "cpu_model": bench.cpu_model(),
               "what": "oracle/rrt_oracle.c, ONE plan at the workload's full size on one thread"}
        out = {"runs": [rec]}
        for d in (os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out")):
            os.makedirs(d, exist_ok=True)
            json.dump(out, open(os.path.join(d, "cpu_fullsize_%s.json" % w), "w"), indent=1)
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
