#!/usr/bin/env python3
"""Aggregate rocprofv3 outputs of the headline bench into the files kept under profiles/.

  python3 tools/pmc_summary.py <dir with *_counter_collection.csv / *_kernel_stats.csv> <tag> \
      --instances 2048 --max-iter 105000 --obstacles 50 [--alg-bytes N]

For every *_counter_collection.csv found below <dir> the counter values are summed per (kernel, counter); the HBM
traffic JSON (profiles/<tag>_traffic.json, read by bench.py) is written when both FETCH_SIZE and WRITE_SIZE passes
are present.  Corrections as prescribed by /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in
KB; on gfx950 FETCH_SIZE reports half of the bytes of 16-B/lane streaming reads, so read bytes = FETCH_SIZE*1024*2;
WRITE_SIZE*1024 is exact.
"""
import argparse
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("tag")
ap.add_argument("--instances", type=int, default=4096)
ap.add_argument("--max-iter", type=int, default=105000)
ap.add_argument("--obstacles", type=int, default=50)
ap.add_argument("--variant", default="q16_mirror")
ap.add_argument("--alg-bytes", type=float, default=None)
ap.add_argument("--workload", default="c2")
ap.add_argument("--kernel-match", default="rrt_star_kernel_v2")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")

sums = defaultdict(float)
disp = defaultdict(set)
for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
            sums[k] += float(row["Counter_Value"])
            disp[k].add((f, row["Dispatch_Id"]))
per_counter = defaultdict(list)
for (k, cn), v in sorted(sums.items()):
    per_counter[cn].append((k, cn, len(disp[(k, cn)]), v))
for cn, rows in per_counter.items():
    out = os.path.join(prof, "%s_pmc_%s.csv" % (a.tag, cn))
    with open(out, "w") as fh:
        fh.write("kernel,counter,dispatches,sum_KB\n")
        for r in rows:
            fh.write("%s,%s,%d,%r\n" % r)
    print("wrote", out)
for f in glob.glob(os.path.join(a.dir, "**", "*kernel_stats.csv"), recursive=True):
    out = os.path.join(prof, "%s_kernel_stats.csv" % a.tag)
    shutil.copy(f, out)
    print("wrote", out)
    break

def dominant(cn):
    rows = [r for r in per_counter.get(cn, []) if a.kernel_match in r[0]]
    return max(rows, key=lambda r: r[3]) if rows else None

fs, wsz = dominant("FETCH_SIZE"), dominant("WRITE_SIZE")
if fs and wsz:
    rd, wr = fs[3] * 1024 * 2, wsz[3] * 1024
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(root, "tools"))
    import csrc_hash as ch
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:  # noqa: BLE001  (the GPU box has no .git: the caller passes the hash through the environment)
        commit = os.environ.get("RRTX_COMMIT", "unknown")
    tj = {"config": {"instances_per_gpu": a.instances, "max_iter": a.max_iter, "obstacles": a.obstacles,
                     "variant": a.variant},
          "csrc_hash": ch.csrc_hash(a.workload), "commit": commit,
          "kernel": fs[0], "dispatches": fs[2], "FETCH_SIZE_KB": fs[3], "WRITE_SIZE_KB": wsz[3],
          "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_step": rd + wr,
          "hbm_bytes_per_launch": (rd + wr) / max(fs[2], 1),
          "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python3 bench.py "
                    "--no-cpu-baseline --warmup 0 --steps 1` (tools/profile_headline.sh); read bytes = "
                    "FETCH_SIZE(KB)*1024*2 (gfx950 correction for 16-B/lane streaming loads, MI355X_MICROARCH.md HBM "
                    "section), write bytes = WRITE_SIZE(KB)*1024"}
    if a.alg_bytes:
        tj["algorithmic_bytes_per_step"] = a.alg_bytes
        tj["traffic_over_algorithmic"] = (rd + wr) / a.alg_bytes
    out = os.path.join(prof, "%s_traffic.json" % a.tag)
    if a.tag[:1] == "r" and a.tag[1:2].isdigit() and a.tag[2:3] == "_":
        out = os.path.join(prof, "%s_%s_traffic.json" % (a.tag[:2], a.workload))
    json.dump(tj, open(out, "w"), indent=1)
    print("wrote", out, json.dumps(tj))
