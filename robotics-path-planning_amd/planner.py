"""Host-side mirrors of the reference planner classes, driving the HIP library.

Same constructor keywords, entry points, attributes and return shapes as
  rrt_01: /root/reference/src_path_planning/10_path_planning_01_rrt_01_simple.py   class RRT :16-101
  rrt_04: /root/reference/src_path_planning/10_path_planning_01_rrt_04_rrt_star.py class RRT :932-1384
so a driver script written against the reference runs unchanged:

    random.seed(s)
    rrt = RRT(start=..., goal=..., obstacle_list=..., rand_area=..., ...)
    path = rrt.planning(animation=False)      # list of [x, y] goal -> start, or None
    len(rrt.node_list); rrt.node_list[i].path_x; rrt.draw_graph()

`planning()` hands CPython's global `random` state to the device (the kernels
consume the MT19937 stream exactly as random.randint / random.uniform would) and
stores the advanced state back, so code after the call sees the same stream it
would have seen with the reference.  stdout chatter of the reference
("Iter: ...", rrt_04:1045) is not reproduced.  All planning runs on the GPU.
"""
import math
import random

import numpy as np

from . import _abi


class Node:
    """RRT.Node (rrt_04:933-942); path_x/path_y are rebuilt on demand for drawing."""

    def __init__(self, x, y):
        self.x = x
        self.y = y
        self.path_x = []
        self.path_y = []
        self.parent = None
        self.cost = 0.0


class AreaBounds:
    """RRT.AreaBounds (rrt_04:944-949)."""

    def __init__(self, area):
        self.xmin = float(area[0])
        self.xmax = float(area[1])
        self.ymin = float(area[2])
        self.ymax = float(area[3])


def _steer_polyline(fx, fy, tx, ty, extend, res):
    """Polyline of steer() (rrt_04:1086-1115) with CPython's own math, for path_x/path_y."""
    nx, ny = fx, fy
    dx, dy = tx - nx, ty - ny
    d, theta = math.hypot(dx, dy), math.atan2(dy, dx)
    px, py = [nx], [ny]
    if extend > d:
        extend = d
    for _ in range(math.floor(extend / res)):
        nx += res * math.cos(theta)
        ny += res * math.sin(theta)
        px.append(nx)
        py.append(ny)
    if math.hypot(tx - nx, ty - ny) <= res:
        px.append(tx)
        py.append(ty)
    return px, py


def _creation_records(trace, kind, n_nodes):
    """(nearest, sample x, sample y, kind) per NODE from the per-ITERATION trace: iterations that appended a node
    (kind 1 / 2) did so in order, node 0 is the start."""
    rx, ry, nearest, _ = trace
    k = min(len(kind), len(rx))
    it = np.nonzero(kind[:k] > 0)[0]
    if len(it) != n_nodes - 1:      # early exit of a launch chunk etc.: no record rather than a wrong one
        return None
    c_near = np.full(n_nodes, -1, dtype=np.int64)
    c_rx = np.zeros(n_nodes)
    c_ry = np.zeros(n_nodes)
    c_kind = np.zeros(n_nodes, dtype=np.int64)
    c_near[1:], c_rx[1:], c_ry[1:], c_kind[1:] = nearest[it], rx[it], ry[it], kind[it]
    return c_near, c_rx, c_ry, c_kind


class NodeList:
    """Lazy `node_list`: SoA arrays from the device, Node objects made on access.

    Parents are object references as in rrt_01/rrt_04 (the same Node object is
    returned for the same index, so identity comparisons behave).

    `path_x` / `path_y` (what draw_graph plots, rrt_04:1165-1167) are rebuilt on the host with CPython's own math, as
    the reference built them.  A node holds the polyline of the steer() call that produced its current entry in
    node_list:
      * re-pointed by a later rewire (its parent index is larger than its own: `node_list[i] = edge_node`, :1372):
        steer(parent -> node, inf) (:1359);
      * appended as the extension itself (rrt_01:85-96; rrt_04:1066-1067 when choose_parent returned None):
        steer(nearest -> sample, expand_dis) (:1051);
      * appended under a chosen parent: steer(parent -> extension end, inf) (:1279).
    `creation` = (nearest index, sample x, sample y, kind) per node, from the device's per-iteration trace
    (rrtx_get_trace / rrtx_get_trace_kind); without it every node falls back to steer(parent -> node, inf).

    Limit (draw-only data): the polylines are rebuilt from the FINAL coordinates of the nearest / parent node.  When a
    later rewire moved that node (an unsnapped steer with an inexact path_resolution, rrt_04:1372), the reference's stored
    polyline still starts at the old position, and a rewired node's path ends at the pre-move target; path_x / path_y are
    exact only while no nearest node or ancestor has moved (with a binary-fraction resolution such as the C2 map's 0.25 none
    moves).  Tree, costs and the returned path are not affected."""

    def __init__(self, x, y, cost, parent, res, expand_dis=None, creation=None):
        self._x, self._y, self._cost, self._parent, self._res = x, y, cost, parent, res
        self._expand_dis, self._creation = expand_dis, creation
        self._cache = {}

    def __len__(self):
        return len(self._x)

    def _polyline(self, j, pj):
        cr = self._creation
        if cr is not None and pj < j and cr[3][j] > 0:
            ne = int(cr[0][j])
            ex, ey = _steer_polyline(float(self._x[ne]), float(self._y[ne]), float(cr[1][j]), float(cr[2][j]),
                                     self._expand_dis, self._res)
            if cr[3][j] == 1:
                return ex, ey
            return _steer_polyline(float(self._x[pj]), float(self._y[pj]), ex[-1], ey[-1], float("inf"), self._res)
        return _steer_polyline(float(self._x[pj]), float(self._y[pj]), float(self._x[j]), float(self._y[j]),
                               float("inf"), self._res)

    def _make(self, i):
        nd = self._cache.get(i)
        if nd is not None:
            return nd
        # iterative parent chain construction (trees are ~100 deep, but avoid recursion limits)
        chain = []
        j = i
        while j >= 0 and j not in self._cache:
            chain.append(j)
            j = int(self._parent[j])
        for j in reversed(chain):
            n = Node(float(self._x[j]), float(self._y[j]))
            n.cost = float(self._cost[j])
            pj = int(self._parent[j])
            if pj >= 0:
                n.parent = self._cache[pj]
                n.path_x, n.path_y = self._polyline(j, pj)
            self._cache[j] = n
        return self._cache[i]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._make(j) for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._make(i)

    def __iter__(self):
        for i in range(len(self)):
            yield self._make(i)


class _PlannerBase:
    Node = Node
    AreaBounds = AreaBounds
    _ALGO = None

    def _common_init(self, start, goal, obstacle_list, rand_area, expand_dis, path_resolution, goal_sample_rate,
                     max_iter, play_area, robot_radius, device):
        self.start = Node(start[0], start[1])
        self.end = Node(goal[0], goal[1])
        self.min_rand = rand_area[0]
        self.max_rand = rand_area[1]
        self.play_area = AreaBounds(play_area) if play_area is not None else None
        self._play_area_arg = play_area
        self.expand_dis = expand_dis
        self.path_resolution = path_resolution
        self.goal_sample_rate = goal_sample_rate
        self.max_iter = max_iter
        self.obstacle_list = obstacle_list
        self.node_list = []
        self.robot_radius = robot_radius
        self.device = device
        self.stats = None
        self._trace = False
        self.trace = None

    def _run(self, sampler, ccd, until_max):
        h = _abi.Handle(self._ALGO, [self.start.x, self.start.y], [self.end.x, self.end.y],
                        [self.min_rand, self.max_rand], self.expand_dis, self.path_resolution, self.goal_sample_rate,
                        self.max_iter, play_area=self._play_area_arg, robot_radius=self.robot_radius, sampler=sampler,
                        connect_circle_dist=ccd, search_until_max_iter=until_max, n_instances=1, device=self.device)
        try:
            h.set_obstacles(self.obstacle_list)
            st = random.getstate()
            h.set_rng_state(0, st)
            h.enable_trace(0)   # per-iteration (sample, nearest, what was appended): Node.path_x / path_y need it
            h.plan(strict=True)
            random.setstate(h.get_rng_state(0, st[2]))
            x, y, cost, parent = h.get_tree(0)
            trace = h.get_trace()
            self.node_list = NodeList(x, y, cost, parent, self.path_resolution, self.expand_dis,
                                      _creation_records(trace, h.get_trace_kind(), len(x)))
            self.tree = (x, y, cost, parent)
            path = h.get_path(0)
            self.stats = h.get_stats()
            if self._trace:
                self.trace = trace
            if sampler == _abi.SAMPLER_SOBOL:
                self.sobol_inter_ = h.get_sobol_index(0)
        finally:
            h.close()
        if path is None:
            return None
        return [[float(px), float(py)] for px, py in path]

    # drawing helpers of the reference (rrt_04:1155-1194); animation happens after the fact
    def draw_graph(self, rnd=None):  # pragma: no cover
        import matplotlib.pyplot as plt
        plt.clf()
        if rnd is not None:
            plt.plot(rnd.x, rnd.y, "^k")
        for node in self.node_list:
            if node.parent:
                plt.plot(node.path_x, node.path_y, "-g")
        for (ox, oy, size) in self.obstacle_list:
            self.plot_circle(ox, oy, size)
        if self.play_area is not None:
            pa = self.play_area
            plt.plot([pa.xmin, pa.xmax, pa.xmax, pa.xmin, pa.xmin], [pa.ymin, pa.ymin, pa.ymax, pa.ymax, pa.ymin], "-k")
        plt.plot(self.start.x, self.start.y, "xr")
        plt.plot(self.end.x, self.end.y, "xr")
        plt.axis("equal")
        plt.grid(True)

    @staticmethod
    def plot_circle(x, y, size, color="-b"):  # pragma: no cover
        import matplotlib.pyplot as plt
        deg = list(range(0, 360, 5)) + [0]
        plt.plot([x + size * math.cos(np.deg2rad(d)) for d in deg],
                 [y + size * math.sin(np.deg2rad(d)) for d in deg], color)

    def calc_dist_to_goal(self, x, y):
        return math.hypot(x - self.end.x, y - self.end.y)


class RRT(_PlannerBase):
    """Drop-in for rrt_01's `RRT` (10_path_planning_01_rrt_01_simple.py:16-101)."""
    _ALGO = _abi.ALGO_RRT

    def __init__(self, start, goal, obstacle_list, rand_area, expand_dis=3.0, path_resolution=0.5,
                 goal_sample_rate=5, max_iter=500, play_area=None, robot_radius=0.0, device=0):
        self._common_init(start, goal, obstacle_list, rand_area, expand_dis, path_resolution, goal_sample_rate,
                          max_iter, play_area, robot_radius, device)

    def planning(self, animation=True):
        path = self._run(_abi.SAMPLER_MT, 50.0, False)
        if animation:  # pragma: no cover
            self.draw_graph()
        return path

    plan = planning


class RRTSobol(RRT):
    """Drop-in for rrt_02's `RRT` (10_path_planning_01_rrt_02_sobol_sampler.py:932-1089): rrt_01 whose
    get_random_node draws the 2-D Sobol sequence (:1077-1089)."""

    def __init__(self, start, goal, obstacle_list, rand_area, expand_dis=3.0, path_resolution=0.5,
                 goal_sample_rate=5, max_iter=500, play_area=None, robot_radius=0.0, device=0):
        super().__init__(start, goal, obstacle_list, rand_area, expand_dis, path_resolution, goal_sample_rate, max_iter,
                         play_area, robot_radius, device)
        self.sobol_inter_ = 0

    def planning(self, animation=True):
        path = self._run(_abi.SAMPLER_SOBOL, 50.0, False)
        if animation:  # pragma: no cover
            self.draw_graph()
        return path

    plan = planning


class RRTStar(_PlannerBase):
    """Drop-in for rrt_04's `RRT` (10_path_planning_01_rrt_04_rrt_star.py:932-1384)."""
    _ALGO = _abi.ALGO_RRT_STAR

    def __init__(self, start, goal, obstacle_list, rand_area, expand_dis=3.0, path_resolution=0.5,
                 goal_sample_rate=5, max_iter=500, play_area=None, robot_radius=0.0, sobol_sampler=True,
                 connect_circle_dist=50.0, search_until_max_iter=False, device=0):
        self._common_init(start, goal, obstacle_list, rand_area, expand_dis, path_resolution, goal_sample_rate,
                          max_iter, play_area, robot_radius, device)
        self.sobol_sampler = sobol_sampler
        self.sobol_inter_ = 0
        self.connect_circle_dist = connect_circle_dist
        self.goal_node = Node(goal[0], goal[1])
        self.search_until_max_iter = search_until_max_iter

    def planning(self, animation=True):
        path = self._run(_abi.SAMPLER_SOBOL if self.sobol_sampler else _abi.SAMPLER_MT, self.connect_circle_dist,
                         self.search_until_max_iter)
        if animation:  # pragma: no cover
            self.draw_graph()
        return path

    plan = planning


class InformedNode:
    """rrt_07's top-level Node (:1020-1025): integer parent index, no path arrays."""

    def __init__(self, x, y):
        self.x = x
        self.y = y
        self.cost = 0.0
        self.parent = None


def informed_rotation(start_xy, goal_xy):
    """c_min and the rotation-to-world matrix `c`, computed with numpy exactly as rrt_07:1054-1068 does."""
    c_min = math.hypot(start_xy[0] - goal_xy[0], start_xy[1] - goal_xy[1])
    a1 = np.array([[(goal_xy[0] - start_xy[0]) / c_min], [(goal_xy[1] - start_xy[1]) / c_min], [0]])
    id1_t = np.array([1.0, 0.0, 0.0]).reshape(1, 3)
    m = a1 @ id1_t
    u, s, vh = np.linalg.svd(m, True, True)
    c = u @ np.diag([1.0, 1.0, np.linalg.det(u) * np.linalg.det(np.transpose(vh))]) @ vh
    return c_min, c


class InformedRRTStar:
    """Drop-in for rrt_07's `RRT` (10_path_planning_01_rrt_07_informed_rrt_star.py:1027-1285)."""

    def __init__(self, start, goal, obstacle_list, rand_area, expand_dis=0.5, goal_sample_rate=10, max_iter=200,
                 sobol_sampler=False, device=0):
        self.start = InformedNode(start[0], start[1])
        self.goal = InformedNode(goal[0], goal[1])
        self.min_rand = rand_area[0]
        self.max_rand = rand_area[1]
        self.expand_dis = expand_dis
        self.goal_sample_rate = goal_sample_rate
        self.max_iter = max_iter
        self.obstacle_list = obstacle_list
        self.node_list = None
        self.sobol_sampler = sobol_sampler
        self.sobol_inter_ = 0
        self.device = device
        self.stats = None
        self._trace = False
        self.trace = None

    def informed_rrt_star_search(self, animation=True):
        c_min, c = informed_rotation([self.start.x, self.start.y], [self.goal.x, self.goal.y])
        h = _abi.Handle(_abi.ALGO_INFORMED, [self.start.x, self.start.y], [self.goal.x, self.goal.y],
                        [self.min_rand, self.max_rand], self.expand_dis, 1.0, self.goal_sample_rate, self.max_iter,
                        sampler=_abi.SAMPLER_SOBOL if self.sobol_sampler else _abi.SAMPLER_MT, n_instances=1,
                        device=self.device, informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
        try:
            h.set_obstacles(self.obstacle_list)
            st = random.getstate()
            h.set_rng_state(0, st)
            if self._trace:
                h.enable_trace(0)
            h.plan(strict=True)
            random.setstate(h.get_rng_state(0, st[2]))
            x, y, cost, parent = h.get_tree(0)
            nodes = []
            for i in range(len(x)):
                nd = InformedNode(float(x[i]), float(y[i]))
                nd.cost = float(cost[i])
                nd.parent = None if parent[i] < 0 else int(parent[i])
                nodes.append(nd)
            self.node_list = nodes
            self.tree = (x, y, cost, parent)
            path = h.get_path(0)
            self.stats = h.get_stats()
            if self._trace:
                self.trace = h.get_trace()
            if self.sobol_sampler:
                self.sobol_inter_ = h.get_sobol_index(0)
        finally:
            h.close()
        return None if path is None else [[float(px), float(py)] for px, py in path]

    plan = informed_rrt_star_search

    @staticmethod
    def get_path_len(path):
        return get_path_length(path)


class DubinsNode:
    """rrt_05's RRT.Node (:1336-1349): pose + the sampled polyline of the edge from its parent."""

    def __init__(self, x, y, yaw):
        self.x = x
        self.y = y
        self.path_x = []
        self.path_y = []
        self.parent = None
        self.cost = 0.0
        self.yaw = yaw
        self.path_yaw = []


class RRTStarDubins:
    """Drop-in for rrt_05's `RRT` (10_path_planning_01_rrt_05_rrt_star_dubins_path.py:1335-1795).

    `planning(animation, search_until_max_iter=True)` as the reference's driver calls it; the sampler is always
    pseudo-random (the reference never calls its Sobol variant, :1426) and edge costs are Euclidean (:1777)."""
    Node = DubinsNode
    AreaBounds = AreaBounds

    def __init__(self, start, goal, obstacle_list, rand_area, expand_dis=3.0, path_resolution=0.5,
                 goal_sample_rate=5, max_iter=500, play_area=None, robot_radius=0.0, sobol_sampler=True,
                 connect_circle_dist=50.0, search_until_max_iter=False, curvature=1.0,
                 goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5, device=0):
        self.start = DubinsNode(start[0], start[1], start[2])
        self.end = DubinsNode(goal[0], goal[1], goal[2])
        self.min_rand = rand_area[0]
        self.max_rand = rand_area[1]
        self.play_area = AreaBounds(play_area) if play_area is not None else None
        self.expand_dis = expand_dis
        self.path_resolution = path_resolution
        self.goal_sample_rate = goal_sample_rate
        self.max_iter = max_iter
        self.obstacle_list = obstacle_list
        self.node_list = []
        self.robot_radius = robot_radius
        self.sobol_sampler = sobol_sampler
        self.sobol_inter_ = 0
        self.connect_circle_dist = connect_circle_dist
        self.search_until_max_iter = search_until_max_iter
        self.curvature = curvature
        self.goal_yaw_th = goal_yaw_th
        self.goal_xy_th = goal_xy_th
        self.device = device
        self.stats = None
        self._trace = False
        self.trace = None

    def planning(self, animation=True, search_until_max_iter=True):
        h = self._make_handle(bool(search_until_max_iter))
        try:
            h.set_obstacles(self.obstacle_list)
            st = random.getstate()
            h.set_rng_state(0, st)
            if self._trace:
                h.enable_trace(0)
            h.plan(strict=True)
            random.setstate(h.get_rng_state(0, st[2]))
            x, y, cost, parent = h.get_tree(0)
            yaw = h.get_yaw(0)
            plen, px, py = h.get_polylines(0)
            nodes, off = [], 0
            for i in range(len(x)):
                nd = DubinsNode(float(x[i]), float(y[i]), float(yaw[i]))
                nd.cost = float(cost[i])
                nd.path_x = px[off:off + plen[i]]
                nd.path_y = py[off:off + plen[i]]
                off += int(plen[i])
                nodes.append(nd)
            for i, nd in enumerate(nodes):
                nd.parent = nodes[int(parent[i])] if parent[i] >= 0 else None
            self.node_list = nodes
            self.tree = (x, y, cost, parent)
            self.yaw = yaw
            self.polylines = (plen, px, py)
            path = h.get_path(0)
            self.stats = h.get_stats()
            self._after_plan(h)
            if self._trace:
                self.trace = h.get_trace()
        finally:
            h.close()
        return None if path is None else [[float(a), float(b)] for a, b in path]

    plan = planning

    def _make_handle(self, until_max=True):
        return _abi.Handle(_abi.ALGO_DUBINS, [self.start.x, self.start.y, self.start.yaw],
                           [self.end.x, self.end.y, self.end.yaw], [self.min_rand, self.max_rand], self.expand_dis,
                           self.path_resolution, self.goal_sample_rate, self.max_iter, robot_radius=self.robot_radius,
                           connect_circle_dist=self.connect_circle_dist, search_until_max_iter=until_max, n_instances=1,
                           device=self.device, curvature=self.curvature, goal_yaw_th=self.goal_yaw_th,
                           goal_xy_th=self.goal_xy_th)

    def _after_plan(self, h):
        pass


class RRTDubins(RRTStarDubins):
    """Drop-in for rrt_03's `RRT` (10_path_planning_01_rrt_03_dubins_path.py:1348-1700): RRT whose steer is the whole
    Dubins path to the sample, node cost = Dubins length (:1458-1481).  `sobol_sampler=True` (the driver's setting)
    draws the 3-D Sobol point of :1545-1563.  As shipped that script stops at import (its word table :1030 names
    functions defined later); the behaviour mirrored here is the one its classes define, pinned by goldens generated
    with that one assignment evaluated after the definitions (oracle/ref_loader.py)."""

    def __init__(self, start, goal, obstacle_list, rand_area, goal_sample_rate=10, max_iter=200, play_area=None,
                 robot_radius=0.0, sobol_sampler=False, curvature=1.0, goal_yaw_th=float(np.deg2rad(1.0)),
                 goal_xy_th=0.5, device=0):
        super().__init__(start, goal, obstacle_list, rand_area, goal_sample_rate=goal_sample_rate, max_iter=max_iter,
                         play_area=play_area, robot_radius=robot_radius, sobol_sampler=sobol_sampler,
                         curvature=curvature, goal_yaw_th=goal_yaw_th, goal_xy_th=goal_xy_th, device=device)

    def _make_handle(self, until_max=True):
        return _abi.Handle(_abi.ALGO_RRT_DUBINS, [self.start.x, self.start.y, self.start.yaw],
                           [self.end.x, self.end.y, self.end.yaw], [self.min_rand, self.max_rand], 0.0, 0.5,
                           self.goal_sample_rate, self.max_iter,
                           play_area=None if self.play_area is None else [self.play_area.xmin, self.play_area.xmax,
                                                                          self.play_area.ymin, self.play_area.ymax],
                           robot_radius=self.robot_radius,
                           sampler=_abi.SAMPLER_SOBOL if self.sobol_sampler else _abi.SAMPLER_MT,
                           search_until_max_iter=until_max, n_instances=1, device=self.device, curvature=self.curvature,
                           goal_yaw_th=self.goal_yaw_th, goal_xy_th=self.goal_xy_th)

    def _after_plan(self, h):
        if self.sobol_sampler:
            self.sobol_inter_ = h.get_sobol_index(0)


class RRTStarReedsShepp(RRTStarDubins):
    """Drop-in for rrt_06's `RRT` (10_path_planning_01_rrt_06_rrt_star_reeds_shepp_path.py:1444-1914): RRT* whose steer
    is the whole Reeds-Shepp path to the sample (:1584-1604), with `try_goal_path` (:1572-1582) after every accepted
    node.  `planning(animation, search_until_max_iter=True)` as the driver calls it (:2087: the constructor's
    `search_until_max_iter` is not read by the loop); the sampler is always `get_random_node` (:1539) and edge costs in
    choose_parent / rewire are Euclidean (the last-defined calc_new_cost, :1901).  Returns the three-column course
    `[[x, y, yaw], ...]` of generate_final_course (:1643-1651)."""

    def __init__(self, start, goal, obstacle_list, rand_area, expand_dis=3.0, path_resolution=0.5,
                 goal_sample_rate=5, max_iter=500, play_area=None, robot_radius=0.0, sobol_sampler=True,
                 connect_circle_dist=50.0, search_until_max_iter=False, curvature=1.0,
                 goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5, step_size=0.2, device=0):
        super().__init__(start, goal, obstacle_list, rand_area, expand_dis=expand_dis, path_resolution=path_resolution,
                         goal_sample_rate=goal_sample_rate, max_iter=max_iter, play_area=play_area,
                         robot_radius=robot_radius, sobol_sampler=sobol_sampler, connect_circle_dist=connect_circle_dist,
                         search_until_max_iter=search_until_max_iter, curvature=curvature, goal_yaw_th=goal_yaw_th,
                         goal_xy_th=goal_xy_th, device=device)
        self.step_size = step_size
        self.path_yaw = None

    def planning(self, animation=True, search_until_max_iter=True):
        self.path_yaw = None
        path = super().planning(animation, search_until_max_iter)
        if path is None:
            return None
        return [[p[0], p[1], float(w)] for p, w in zip(path, self.path_yaw)]

    plan = planning

    def _make_handle(self, until_max=True):
        return _abi.Handle(_abi.ALGO_RS, [self.start.x, self.start.y, self.start.yaw],
                           [self.end.x, self.end.y, self.end.yaw], [self.min_rand, self.max_rand], self.expand_dis,
                           self.path_resolution, self.goal_sample_rate, self.max_iter, robot_radius=self.robot_radius,
                           connect_circle_dist=self.connect_circle_dist, search_until_max_iter=until_max, n_instances=1,
                           device=self.device, curvature=self.curvature, goal_yaw_th=self.goal_yaw_th,
                           goal_xy_th=self.goal_xy_th, step_size=self.step_size)

    def _after_plan(self, h):
        self.path_yaw = h.get_path_yaw(0)


def path_smoothing(path, max_iter, obstacle_list, device=0):
    """Drop-in for rrt_04's module function `path_smoothing(path, max_iter, obstacle_list)` (:1447-1479): random
    shortcutting of the path `planning()` returned, drawing from CPython's global `random` stream (left exactly where
    the reference would leave it).  Runs on the GPU through the C ABI; see `Handle.smooth_planned` for batches."""
    st = random.getstate()
    out, states, _ = _abi.smooth_paths([path], max_iter, obstacle_list, [(np.array(st[1][:624], dtype=np.uint32), st[1][624])],
                                       device=device)
    w, pos = states[0]
    random.setstate((st[0], tuple(int(v) for v in w) + (int(pos),), st[2]))
    return [[float(a), float(b)] for a, b in out[0]]


def bitstar_rotation(start_xy, goal_xy):
    """cMin and C of rrt_08:189-202, computed with numpy exactly as the reference does."""
    c_min = math.hypot(start_xy[0] - goal_xy[0], start_xy[1] - goal_xy[1]) / 1.5
    a1 = np.array([[(goal_xy[0] - start_xy[0]) / c_min], [(goal_xy[1] - start_xy[1]) / c_min], [0]])
    id1_t = np.array([1.0, 0.0, 0.0]).reshape(1, 3)
    m = np.dot(a1, id1_t)
    u, s, vh = np.linalg.svd(m, True, True)
    c = np.dot(np.dot(u, np.diag([1.0, 1.0, np.linalg.det(u) * np.linalg.det(np.transpose(vh))])), vh)
    return c_min, c


class BITStar:
    """Drop-in for rrt_08's `BITStar` (10_path_planning_01_rrt_08_batch_informed_rrt_star.py:138-566).
    As in the reference, `lowerLimit`, `upperLimit`, `resolution` and `eta` are accepted and ignored (:165-168)."""

    def __init__(self, start, goal, obstacleList, randArea, eta=2.0, maxIter=80, lowerLimit=None, upperLimit=None,
                 resolution=0.01, device=0):
        self.start = start
        self.goal = goal
        self.min_rand = randArea[0]
        self.max_rand = randArea[1]
        self.max_iIter = maxIter
        self.obstacleList = obstacleList
        self.eta = eta
        self.device = device
        self.stats = None
        self.tree_arrays = None
        self._trace = False
        self.trace = None

    def plan(self, animation=True):
        c_min, c = bitstar_rotation(self.start, self.goal)
        h = _abi.Handle(_abi.ALGO_BITSTAR, [self.start[0], self.start[1]], [self.goal[0], self.goal[1]],
                        [self.min_rand, self.max_rand], 2.0, 1.0, 0, self.max_iIter, n_instances=1, device=self.device,
                        informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
        try:
            h.set_obstacles(self.obstacleList)
            st = random.getstate()
            h.set_rng_state(0, st)
            if self._trace:
                h.enable_trace(0)
            h.plan(strict=True)
            random.setstate(h.get_rng_state(0, st[2]))
            self.tree_arrays = h.get_tree(0)
            path = h.get_path(0)
            self.stats = h.get_stats()
            if self._trace:
                self.trace = h.get_trace()
        finally:
            h.close()
        return [] if path is None else [[float(a), float(b)] for a, b in path]


def get_path_length(path):
    """rrt_04:1391-1399."""
    le = 0
    for i in range(len(path) - 1):
        le += math.hypot(path[i + 1][0] - path[i][0], path[i + 1][1] - path[i][1])
    return le


class BatchPlanner:
    """Many independent planning instances (seeds / start-goal pairs) on one GPU or sharded over several.

    This is the throughput form of the same kernels: instance i consumes the stream of
    `random.seed(seeds[i])` and produces exactly the tree the single-instance class would.

    `devices=[d0, d1, ...]` ("one handle per device, driven from one process with N threads"): the batch is cut
    into len(devices) contiguous blocks (`sharding.split_contiguous`), one handle per entry, planned concurrently by
    `rrtx_plan_many` (one native host thread per handle; no torch, no collective -- instances never communicate) and
    read back as ONE batch: every accessor takes the global instance index.  A device may be listed more than once
    (two handles sharing a GPU), which is also how the sharding is tested on a one-GPU box.  Results do not depend on
    the sharding."""

    def __init__(self, algo, seeds, start, goal, obstacle_list, rand_area, expand_dis=3.0, path_resolution=0.5,
                 goal_sample_rate=5, max_iter=500, play_area=None, robot_radius=0.0, sobol_sampler=False,
                 connect_circle_dist=50.0, search_until_max_iter=False, device=0, starts=None, goals=None,
                 curvature=1.0, goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5, step_size=0.2, devices=None):
        """algo: "rrt" (rrt_01/02), "rrt_star" (rrt_04), "informed" (rrt_07: expand_dis, goal_sample_rate, max_iter,
        sobol_sampler as in its constructor :1029-1042), "bitstar" (rrt_08: max_iter = maxIter, rand_area = randArea
        :140-168), and the pose planners (start / goal = [x, y, yaw]; curvature, goal thresholds and, for Reeds-Shepp,
        step_size as in their constructors): "rrt_dubins" (rrt_03), "rrt_star_dubins" (rrt_05),
        "rrt_star_reeds_shepp" (rrt_06).
        `starts` / `goals`: per-instance [x, y] (pose planners: [x, y, yaw]; a missing yaw keeps `start[2]` /
        `goal[2]`).  For "informed" and "bitstar" the rotation to the world frame and c_min (rrt_07:1054-1068,
        rrt_08:189-202) are computed per instance on the host with numpy, as the reference does per planner object.
        `devices`: HIP device ordinals to shard over (default: [device])."""
        from . import sharding
        a = {"rrt": _abi.ALGO_RRT, "rrt_star": _abi.ALGO_RRT_STAR, "rrt_dubins": _abi.ALGO_RRT_DUBINS,
             "rrt_star_dubins": _abi.ALGO_DUBINS, "rrt_star_reeds_shepp": _abi.ALGO_RS, "informed": _abi.ALGO_INFORMED,
             "bitstar": _abi.ALGO_BITSTAR}[algo]
        self.seeds = list(seeds)
        self.pose = a in (_abi.ALGO_RRT_DUBINS, _abi.ALGO_DUBINS, _abi.ALGO_RS)
        self.algo = a
        n = len(self.seeds)
        self.devices = [int(device)] if devices is None else [int(d) for d in devices]
        if not self.devices or n < len(self.devices):
            raise ValueError("BatchPlanner: %d instances cannot be sharded over %d handles" % (n, len(self.devices)))
        self.shards = sharding.split_contiguous(n, len(self.devices))
        rot_of = {_abi.ALGO_INFORMED: informed_rotation, _abi.ALGO_BITSTAR: bitstar_rotation}.get(a)
        self.handles = []
        try:
            for dev, (lo, hi) in zip(self.devices, self.shards):
                if a == _abi.ALGO_INFORMED:
                    c_min, c = informed_rotation(start, goal)
                    h = _abi.Handle(a, start, goal, rand_area, expand_dis, 1.0, goal_sample_rate, max_iter,
                                    sampler=_abi.SAMPLER_SOBOL if sobol_sampler else _abi.SAMPLER_MT, n_instances=hi - lo,
                                    device=dev, informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
                elif a == _abi.ALGO_BITSTAR:
                    c_min, c = bitstar_rotation(start, goal)
                    h = _abi.Handle(a, start, goal, rand_area, 2.0, 1.0, 0, max_iter, n_instances=hi - lo, device=dev,
                                    informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
                else:
                    h = _abi.Handle(a, start, goal, rand_area, expand_dis, path_resolution, goal_sample_rate, max_iter,
                                    play_area=play_area, robot_radius=robot_radius,
                                    sampler=_abi.SAMPLER_SOBOL if sobol_sampler else _abi.SAMPLER_MT,
                                    connect_circle_dist=connect_circle_dist, search_until_max_iter=search_until_max_iter,
                                    n_instances=hi - lo, device=dev, curvature=curvature, goal_yaw_th=goal_yaw_th,
                                    goal_xy_th=goal_xy_th, step_size=step_size)
                self.handles.append(h)
                h.set_obstacles(obstacle_list)
                h.seed_instances(self.seeds[lo:hi])
                if starts is not None or goals is not None:
                    for i in range(lo, hi):
                        si = start if starts is None else starts[i]
                        gi = goal if goals is None else goals[i]
                        h.set_instance(i - lo, None if starts is None else si, None if goals is None else gi)
                        if rot_of is not None:
                            cm, ci = rot_of(si, gi)
                            h.set_instance_rotation(i - lo, [ci[0, 0], ci[0, 1], ci[1, 0], ci[1, 1]], cm)
        except Exception:
            self.close()
            raise
        self.h = self.handles[0]      # the single-device form's handle (kept for callers that reach for it)
        self.partial = False

    def _loc(self, i):
        """(handle, local index) of global instance i."""
        if i < 0:
            i += len(self.seeds)
        for h, (lo, hi) in zip(self.handles, self.shards):
            if lo <= i < hi:
                return h, i - lo
        raise IndexError(i)

    def plan(self):
        """Plans every instance; returns (path_cost, n_nodes, status) per instance.  An instance that stopped on a
        capacity limit or where the reference would raise carries the bit in its status word (`failed()` lists
        them); the other instances are complete.  Raises only for errors of the call as a whole."""
        rcs = _abi.plan_many(self.handles)
        self.partial = any(r == _abi.RRTX_PARTIAL for r in rcs)
        return self.results()

    def results(self):
        """(path_cost, n_nodes, status) of the whole batch, shards concatenated in instance order."""
        parts = [h.get_results() for h in self.handles]
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))

    def failed(self):
        """Indices of the instances without a result after plan(), with their status words."""
        _, _, st = self.results()
        return [(int(i), int(st[i])) for i in np.nonzero(st & _abi.ST_FAILED)[0]]

    def stats(self):
        """rrtx_stats summed over the shards (times: the slowest shard's; maxima: the largest)."""
        per = [h.get_stats() for h in self.handles]
        if len(per) == 1:
            return per[0]
        out = {}
        for k in per[0]:
            v = [p[k] for p in per]
            out[k] = max(v) if k in ("kernel_ms", "plan_ms", "kernel_ms_main", "near_unique_max", "main_shape", "main_f32") else sum(v)
        out["per_shard"] = per
        return out

    def tree(self, i):
        h, j = self._loc(i)
        return h.get_tree(j)

    def path(self, i):
        """The course `planning()` returns: (n, 2), or (n, 3) with the yaw column for "rrt_star_reeds_shepp"."""
        h, j = self._loc(i)
        p = h.get_path(j)
        if p is not None and self.algo == _abi.ALGO_RS:
            p = np.column_stack([p, h.get_path_yaw(j)])
        return p

    def yaw(self, i):
        h, j = self._loc(i)
        return h.get_yaw(j)

    def polylines(self, i):
        h, j = self._loc(i)
        return h.get_polylines(j)

    def rng_state(self, i):
        h, j = self._loc(i)
        return h.get_rng_state(j)

    def smooth(self, max_iter):
        """path_smoothing(path, max_iter, obstacle_list) (rrt_04:1447-1479) on every planned path, on the device, each
        instance continuing its own random stream (as the driver does at :1558-1559)."""
        for h in self.handles:
            h.smooth_planned(max_iter)
        return [self._loc(i)[0].get_smoothed_path(self._loc(i)[1]) for i in range(len(self.seeds))]

    def export_npz(self, filename, instances=None):
        """Compact on-disk form of the planned trees for plotting / regression diffs: per instance
        (x, y, cost, parent) as the reference's node_list holds them, the returned path, path cost, seed."""
        ids = list(range(len(self.seeds))) if instances is None else list(instances)
        pc, nn, st = self.results()
        out = dict(seeds=np.array([self.seeds[i] for i in ids], dtype=np.int64), path_cost=pc[ids], n_nodes=nn[ids],
                   status=st[ids])
        for k, i in enumerate(ids):
            h, j = self._loc(i)
            x, y, cost, parent = h.get_tree(j)
            p = h.get_path(j)
            out["x_%d" % k], out["y_%d" % k], out["cost_%d" % k], out["parent_%d" % k] = x, y, cost, parent
            out["path_%d" % k] = np.zeros((0, 2)) if p is None else p
        np.savez_compressed(filename, **out)
        return filename

    def close(self):
        for h in getattr(self, "handles", []):
            h.close()
