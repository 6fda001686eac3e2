"""Names of /root/reference/src_path_planning/10_path_planning_01_rrt_08_batch_informed_rrt_star.py as its driver cell uses them: BITStar :138-566.
Each is the MI355X mirror class / function of robotics-path-planning_amd/planner.py (same constructor keywords and
defaults, same entry points and return shapes)."""
from . import planner as _p

BITStar = _p.BITStar

__all__ = ['BITStar']
