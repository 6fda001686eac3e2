"""Names of /root/reference/src_path_planning/10_path_planning_01_rrt_04_rrt_star.py as its driver cell uses them: RRT :932-1384 (RRT*), get_path_length :1391, path_smoothing :1447.
Each is the MI355X mirror class / function of robotics-path-planning_amd/planner.py (same constructor keywords and
defaults, same entry points and return shapes)."""
from . import planner as _p

RRT = _p.RRTStar
get_path_length = _p.get_path_length
path_smoothing = _p.path_smoothing

__all__ = ['RRT', 'get_path_length', 'path_smoothing']
