"""Names of /root/reference/src_path_planning/10_path_planning_01_rrt_07_informed_rrt_star.py as its driver cell uses them: Node :1020-1025, RRT :1027-1285 (Informed RRT*).
Each is the MI355X mirror class / function of robotics-path-planning_amd/planner.py (same constructor keywords and
defaults, same entry points and return shapes)."""
from . import planner as _p

RRT = _p.InformedRRTStar
Node = _p.InformedNode

__all__ = ['RRT', 'Node']
