"""MI355X-native batched RRT / RRT* planner (host mirror of the reference classes).

The directory name carries a hyphen (it mirrors the upstream repository name), so
import it with importlib, or through the `rrt_amd` shim at the repository root:

    import rrt_amd                       # == importlib.import_module("robotics-path-planning_amd")
    rrt = rrt_amd.RRTStar(start, goal, obstacle_list, rand_area, ...)
    path = rrt.planning(animation=False)

Submodules: planner (RRT = rrt_01's class, RRTStar = rrt_04's class, BatchPlanner),
_abi (ctypes binding of include/rrtx.h), csrc/ (HIP kernels + C ABI sources).
"""
from . import _abi  # noqa: F401
from .planner import (RRT, RRTSobol, RRTStar, RRTStarDubins, RRTDubins, RRTStarReedsShepp, BITStar, bitstar_rotation, InformedRRTStar, BatchPlanner, Node, AreaBounds, get_path_length, path_smoothing,  # noqa: F401
                      informed_rotation)

__all__ = ["RRT", "RRTSobol", "RRTStar", "RRTStarDubins", "RRTDubins", "RRTStarReedsShepp", "BITStar", "InformedRRTStar", "BatchPlanner", "Node", "AreaBounds", "get_path_length", "path_smoothing"]
