#!/usr/bin/env python3
"""Identity of the device code a measurement belongs to: sha256 over the source files the workload's kernel is built
from.  profiles/r2_*_traffic.json / r2_*_valu.json record it; bench.py reports a PMC figure only when it matches the
code it is running (a profile of other code is never rescaled onto a new kernel).
  python3 tools/csrc_hash.py c2"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["rrtx_api.hip", "rrt_kernels.hip.h", "rpp_core.h", "glibc235_fma_math.h"]
FILES = {
    "c2": COMMON + ["rrt_star_v2.hip.h", "rrt_star_v2_body.inc"],
    "c3": COMMON + ["rrt_informed.hip.h"],
    "c4": COMMON + ["rrt_bitstar.hip.h", "rrt_bitstar_wave.hip.h", "rpp_bitstar.h"],
    "c5": COMMON + ["rrt_dubins.hip.h", "rpp_dubins.h"],
    "c6": COMMON + ["rrt_rs.hip.h", "rpp_rs.h", "rrt_dubins.hip.h", "rpp_dubins.h"],
}


def csrc_hash(workload="c2"):
    h = hashlib.sha256()
    d = os.path.join(ROOT, "robotics-path-planning_amd", "csrc")
    for f in sorted(FILES[workload]):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash(sys.argv[1] if len(sys.argv) > 1 else "c2"))
