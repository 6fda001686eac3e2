"""Names of /root/reference/src_path_planning/10_path_planning_01_rrt_06_rrt_star_reeds_shepp_path.py as its driver cell uses them: RRT :1444-1914 (RRT*-Reeds-Shepp).
Each is the MI355X mirror class / function of robotics-path-planning_amd/planner.py (same constructor keywords and
defaults, same entry points and return shapes)."""
from . import planner as _p

RRT = _p.RRTStarReedsShepp

__all__ = ['RRT']
