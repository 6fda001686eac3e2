"""Importable alias of the `robotics-path-planning_amd` package (hyphenated directory name).

Per-script drop-in modules: `rrt_amd.rrt_01` ... `rrt_amd.rrt_08` carry exactly the names the reference script of that
number defines for its driver cell (`RRT`, `BITStar`, `Node`, `path_smoothing`, `get_path_length`), so a driver written
against `10_path_planning_01_rrt_04_rrt_star.py` runs after `from rrt_amd.rrt_04 import *`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("robotics-path-planning_amd")
_abi = _pkg._abi
planner = importlib.import_module("robotics-path-planning_amd.planner")
RRT = _pkg.RRT
RRTStar = _pkg.RRTStar
RRTSobol = _pkg.RRTSobol
RRTStarDubins = _pkg.RRTStarDubins
RRTDubins = _pkg.RRTDubins
RRTStarReedsShepp = _pkg.RRTStarReedsShepp
path_smoothing = _pkg.path_smoothing
BITStar = _pkg.BITStar
bitstar_rotation = _pkg.bitstar_rotation
InformedRRTStar = _pkg.InformedRRTStar
informed_rotation = _pkg.informed_rotation
BatchPlanner = _pkg.BatchPlanner
Node = _pkg.Node
AreaBounds = _pkg.AreaBounds
get_path_length = _pkg.get_path_length
InformedNode = planner.InformedNode
DubinsNode = planner.DubinsNode
