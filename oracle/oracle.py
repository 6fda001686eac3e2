"""ctypes wrapper of the CPU oracle (oracle/rrt_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(C.Structure):
    _fields_ = [("algo", C.c_int32), ("goal_sample_rate", C.c_int32), ("max_iter", C.c_int32),
                ("has_play_area", C.c_int32), ("sobol", C.c_int32), ("search_until_max_iter", C.c_int32),
                ("exact_pow", C.c_int32), ("pad_", C.c_int32),
                ("start", C.c_double * 2), ("goal", C.c_double * 2),
                ("rand_min", C.c_double), ("rand_max", C.c_double),
                ("expand_dis", C.c_double), ("path_resolution", C.c_double),
                ("play_area", C.c_double * 4), ("robot_radius", C.c_double),
                ("connect_circle_dist", C.c_double)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("edges_ref", "edges_unique", "near_hits", "near_unique", "rewires",
                                         "propagated", "iterations", "scan_nodes", "pow_slow", "sobol_index")]


class MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("pos", C.c_int32)]


class Out(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("cost", C.c_void_p), ("parent", C.c_void_p),
                ("cap", C.c_int32), ("n", C.c_int32),
                ("path_xy", C.c_void_p), ("path_cap", C.c_int32), ("path_n", C.c_int32),
                ("tr_rnd_x", C.c_void_p), ("tr_rnd_y", C.c_void_p), ("tr_nearest", C.c_void_p),
                ("tr_n_near", C.c_void_p), ("tr_cap", C.c_int32), ("tr_n", C.c_int32), ("path_yaw", C.c_void_p)]


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "rrt_oracle.c")):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_plan.restype = C.c_int
        _LIB.orc_hypot.restype = C.c_double
        _LIB.orc_hypot.argtypes = [C.c_double, C.c_double]
    return _LIB


def mt_from_seed(seed):
    s = MT()
    lib().orc_mt_seed(C.byref(s), C.c_uint64(abs(int(seed))))
    return s


def mt_from_pystate(state):
    """state = random.getstate()"""
    s = MT()
    for i in range(624):
        s.mt[i] = state[1][i]
    s.pos = state[1][624]
    return s


def plan(algo="rrt_star", start=(0, 0), goal=(6, 10), obstacles=(), rand_area=(-2, 15), expand_dis=3.0,
         path_resolution=0.5, goal_sample_rate=5, max_iter=500, play_area=None, robot_radius=0.0, sobol=False,
         connect_circle_dist=50.0, search_until_max_iter=False, seed=None, rng=None, exact_pow=True, trace=False):
    """Run one planning() call on the oracle; returns dict(x,y,cost,parent,path,stats,rng,trace...)."""
    L = lib()
    p = Params()
    p.algo = {"rrt": 0, "rrt_star": 1}[algo]
    p.goal_sample_rate = int(goal_sample_rate)
    p.max_iter = int(max_iter)
    p.has_play_area = 0 if play_area is None or len(play_area) == 0 else 1
    p.sobol = int(bool(sobol))
    p.search_until_max_iter = int(bool(search_until_max_iter))
    p.exact_pow = int(bool(exact_pow))
    p.start[0], p.start[1] = float(start[0]), float(start[1])
    p.goal[0], p.goal[1] = float(goal[0]), float(goal[1])
    p.rand_min, p.rand_max = float(rand_area[0]), float(rand_area[1])
    p.expand_dis, p.path_resolution = float(expand_dis), float(path_resolution)
    if p.has_play_area:
        for i in range(4):
            p.play_area[i] = float(play_area[i])
    p.robot_radius = float(robot_radius)
    p.connect_circle_dist = float(connect_circle_dist)
    obst = np.ascontiguousarray(np.array(obstacles, dtype=np.float64).reshape(-1, 3))
    if rng is None:
        rng = mt_from_seed(seed)
    cap = int(max_iter) + 2
    x = np.zeros(cap); y = np.zeros(cap); cost = np.zeros(cap); parent = np.zeros(cap, dtype=np.int32)
    path = np.zeros((cap + 2, 2))
    o = Out()
    o.x, o.y, o.cost, o.parent = x.ctypes.data, y.ctypes.data, cost.ctypes.data, parent.ctypes.data
    o.cap = cap
    o.path_xy, o.path_cap = path.ctypes.data, cap + 2
    if trace:
        trx = np.zeros(max_iter); try_ = np.zeros(max_iter)
        trn = np.zeros(max_iter, dtype=np.int32); trk = np.zeros(max_iter, dtype=np.int32)
        o.tr_rnd_x, o.tr_rnd_y, o.tr_nearest, o.tr_n_near = trx.ctypes.data, try_.ctypes.data, trn.ctypes.data, trk.ctypes.data
        o.tr_cap = int(max_iter)
    st = Stats()
    rc = L.orc_plan(C.byref(p), obst.ctypes.data_as(C.c_void_p), C.c_int(len(obst)), C.byref(rng), C.byref(o),
                    C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_plan failed: %d" % rc)
    n = o.n
    res = dict(x=x[:n].copy(), y=y[:n].copy(), cost=cost[:n].copy(), parent=parent[:n].copy(),
               path=path[:o.path_n].copy() if o.path_n else None,
               stats={k: getattr(st, k) for k, _ in Stats._fields_}, rng=rng)
    if trace:
        t = o.tr_n
        res.update(tr_rnd_x=trx[:t].copy(), tr_rnd_y=try_[:t].copy(), tr_nearest=trn[:t].copy(), tr_n_near=trk[:t].copy())
    return res


def rotation_to_world(start, goal):
    """The matrix `c` of rrt_07:1054-1068, computed with numpy exactly as the reference does (SVD of a1 . e1^T)."""
    import math
    c_min = math.hypot(start[0] - goal[0], start[1] - goal[1])
    a1 = np.array([[(goal[0] - start[0]) / c_min], [(goal[1] - start[1]) / c_min], [0]])
    id1_t = np.array([1.0, 0.0, 0.0]).reshape(1, 3)
    m = a1 @ id1_t
    u, s, vh = np.linalg.svd(m, True, True)
    return u @ np.diag([1.0, 1.0, np.linalg.det(u) * np.linalg.det(np.transpose(vh))]) @ vh


def plan_informed(start, goal, obstacles, rand_area, expand_dis=0.5, goal_sample_rate=10, max_iter=200, sobol=False,
                  seed=None, rng=None, exact_pow=True, trace=False, rot_c=None):
    """One RRT.informed_rrt_star_search() call of rrt_07 on the oracle."""
    L = lib()
    L.orc_plan_informed.restype = C.c_int
    p = Params()
    p.algo = 2
    p.goal_sample_rate, p.max_iter = int(goal_sample_rate), int(max_iter)
    p.sobol, p.exact_pow = int(bool(sobol)), int(bool(exact_pow))
    p.start[0], p.start[1] = float(start[0]), float(start[1])
    p.goal[0], p.goal[1] = float(goal[0]), float(goal[1])
    p.rand_min, p.rand_max = float(rand_area[0]), float(rand_area[1])
    p.expand_dis = float(expand_dis)
    obst = np.ascontiguousarray(np.array(obstacles, dtype=np.float64).reshape(-1, 3))
    if rng is None:
        rng = mt_from_seed(seed)
    if rot_c is None:
        rot_c = rotation_to_world([float(start[0]), float(start[1])], [float(goal[0]), float(goal[1])])
    rc_ = np.ascontiguousarray(np.array(rot_c, dtype=np.float64).reshape(9))
    cap = int(max_iter) + 2
    x = np.zeros(cap); y = np.zeros(cap); cost = np.zeros(cap); parent = np.zeros(cap, dtype=np.int32)
    path = np.zeros((cap + 2, 2))
    o = Out()
    o.x, o.y, o.cost, o.parent = x.ctypes.data, y.ctypes.data, cost.ctypes.data, parent.ctypes.data
    o.cap = cap
    o.path_xy, o.path_cap = path.ctypes.data, cap + 2
    if trace:
        trx = np.zeros(max_iter); try_ = np.zeros(max_iter)
        trn = np.zeros(max_iter, dtype=np.int32); trk = np.zeros(max_iter, dtype=np.int32)
        o.tr_rnd_x, o.tr_rnd_y, o.tr_nearest, o.tr_n_near = trx.ctypes.data, try_.ctypes.data, trn.ctypes.data, trk.ctypes.data
        o.tr_cap = int(max_iter)
    st = Stats()
    cb = C.c_double()
    rc = L.orc_plan_informed(C.byref(p), obst.ctypes.data_as(C.c_void_p), C.c_int(len(obst)),
                             rc_.ctypes.data_as(C.c_void_p), C.byref(rng), C.byref(o), C.byref(st), C.byref(cb))
    if rc != 0:
        raise RuntimeError("orc_plan_informed failed: %d" % rc)
    n = o.n
    res = dict(x=x[:n].copy(), y=y[:n].copy(), cost=cost[:n].copy(), parent=parent[:n].copy(),
               path=path[:o.path_n].copy() if o.path_n else None, c_best=cb.value,
               stats={k: getattr(st, k) for k, _ in Stats._fields_}, rng=rng)
    if trace:
        t = o.tr_n
        res.update(tr_rnd_x=trx[:t].copy(), tr_rnd_y=try_[:t].copy(), tr_nearest=trn[:t].copy(), tr_n_near=trk[:t].copy())
    return res


class DOut(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("yaw", C.c_void_p), ("cost", C.c_void_p), ("parent", C.c_void_p),
                ("cap", C.c_int32), ("n", C.c_int32),
                ("poly_len", C.c_void_p), ("poly_x", C.c_void_p), ("poly_y", C.c_void_p),
                ("poly_cap", C.c_int64), ("poly_n", C.c_int64),
                ("path_xy", C.c_void_p), ("path_cap", C.c_int32), ("path_n", C.c_int32),
                ("tr_rx", C.c_void_p), ("tr_ry", C.c_void_p), ("tr_ryaw", C.c_void_p), ("tr_nearest", C.c_void_p),
                ("tr_n_near", C.c_void_p), ("tr_cap", C.c_int32), ("tr_n", C.c_int32), ("path_yaw", C.c_void_p)]


def dubins(sx, sy, syaw, gx, gy, gyaw, curvature=1.0, cap=4096):
    """plan_dubins_path (rrt_05:1021-1109) on the oracle: (px, py, pyaw, mode, lengths)."""
    L = lib()
    L.orc_dubins.restype = C.c_int
    L.orc_dubins.argtypes = [C.c_double] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p]
    px = np.zeros(cap); py = np.zeros(cap); pyaw = np.zeros(cap); ln = np.zeros(3)
    mode = C.create_string_buffer(4)
    n = L.orc_dubins(sx, sy, syaw, gx, gy, gyaw, curvature, px.ctypes.data, py.ctypes.data, pyaw.ctypes.data, cap,
                     ln.ctypes.data, mode)
    return px[:n].copy(), py[:n].copy(), pyaw[:n].copy(), mode.value.decode(), ln


def plan_rrt_dubins(start, goal, obstacles, rand_area, max_iter=200, seed=None, rng=None, curvature=1.0,
                    robot_radius=0.0, goal_sample_rate=10, goal_yaw_th=None, goal_xy_th=0.5, sobol=False, play_area=None,
                    trace=False, search_until_max_iter=True):
    """One RRT.planning(animation=False) call of rrt_03 (RRT with Dubins steer) on the oracle."""
    return plan_dubins(start, goal, obstacles, rand_area, max_iter, seed=seed, rng=rng, curvature=curvature,
                       robot_radius=robot_radius, goal_sample_rate=goal_sample_rate, goal_yaw_th=goal_yaw_th,
                       goal_xy_th=goal_xy_th, trace=trace, search_until_max_iter=search_until_max_iter, _plain=True,
                       _sobol=sobol, _play_area=play_area)


def set_lazy_order(on):
    """Test switch of rrt_oracle.c (orc_set_lazy_order): candidate order of the device's pose-tree kernels."""
    L = lib()
    L.orc_set_lazy_order.argtypes = [C.c_int]
    L.orc_set_lazy_order.restype = None
    L.orc_set_lazy_order(int(bool(on)))


def plan_rrt_rs(start, goal, obstacles, rand_area, max_iter=500, seed=None, rng=None, curvature=1.0, robot_radius=0.0,
                expand_dis=3.0, connect_circle_dist=50.0, goal_yaw_th=None, goal_xy_th=0.5, step_size=0.2, trace=False,
                search_until_max_iter=True):
    """One RRT.planning(animation=False) call of rrt_06 (RRT*-Reeds-Shepp, :1530-1570) on the oracle.  The returned
    dict's `path` / `path_yaw` are the three columns of generate_final_course (:1643-1651)."""
    return plan_dubins(start, goal, obstacles, rand_area, max_iter, seed=seed, rng=rng, curvature=curvature,
                       robot_radius=robot_radius, expand_dis=expand_dis, connect_circle_dist=connect_circle_dist,
                       goal_yaw_th=goal_yaw_th, goal_xy_th=goal_xy_th, trace=trace,
                       search_until_max_iter=search_until_max_iter, _rs_step=float(step_size))


def plan_dubins(start, goal, obstacles, rand_area, max_iter=500, seed=None, rng=None, curvature=1.0, robot_radius=0.0,
                goal_sample_rate=10, expand_dis=3.0, connect_circle_dist=50.0, goal_yaw_th=None, goal_xy_th=0.5,
                trace=False, search_until_max_iter=True, _plain=False, _sobol=False, _play_area=None, _rs_step=0.0):
    """One RRT.planning(animation=False) call of rrt_05 (RRT*-Dubins) on the oracle."""
    L = lib()
    L.orc_plan_rrt_rs.restype = C.c_int
    L.orc_plan_rrt_rs.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_plan_dubins.restype = C.c_int
    L.orc_plan_dubins.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_plan_rrt_dubins.restype = C.c_int
    L.orc_plan_rrt_dubins.argtypes = L.orc_plan_dubins.argtypes + [C.c_void_p]
    p = Params()
    p.algo = 3
    p.sobol = int(bool(_sobol))
    p.search_until_max_iter = int(bool(search_until_max_iter))
    if _play_area is not None:
        p.has_play_area = 1
        for k in range(4):
            p.play_area[k] = float(_play_area[k])
    p.goal_sample_rate, p.max_iter = int(goal_sample_rate), int(max_iter)
    p.start[0], p.start[1] = float(start[0]), float(start[1])
    p.goal[0], p.goal[1] = float(goal[0]), float(goal[1])
    p.rand_min, p.rand_max = float(rand_area[0]), float(rand_area[1])
    p.expand_dis = float(expand_dis)
    p.robot_radius = float(robot_radius)
    p.connect_circle_dist = float(connect_circle_dist)
    if goal_yaw_th is None:
        goal_yaw_th = float(np.deg2rad(1.0))
    obst = np.ascontiguousarray(np.array(obstacles, dtype=np.float64).reshape(-1, 3))
    if rng is None:
        rng = mt_from_seed(seed)
    cap = (2 if _rs_step > 0 else 1) * int(max_iter) + 2
    pcap = cap * 2048
    x = np.zeros(cap); y = np.zeros(cap); yaw = np.zeros(cap); cost = np.zeros(cap); parent = np.zeros(cap, dtype=np.int32)
    plen = np.zeros(cap, dtype=np.int32); ppx = np.zeros(pcap); ppy = np.zeros(pcap)
    path = np.zeros((pcap, 2)); path_yaw = np.zeros(pcap)
    o = DOut()
    o.path_yaw = path_yaw.ctypes.data
    o.x, o.y, o.yaw, o.cost, o.parent = x.ctypes.data, y.ctypes.data, yaw.ctypes.data, cost.ctypes.data, parent.ctypes.data
    o.cap = cap
    o.poly_len, o.poly_x, o.poly_y, o.poly_cap = plen.ctypes.data, ppx.ctypes.data, ppy.ctypes.data, pcap
    o.path_xy, o.path_cap = path.ctypes.data, pcap
    if trace:
        trx = np.zeros(max_iter); try_ = np.zeros(max_iter); tryaw = np.zeros(max_iter)
        trn = np.zeros(max_iter, dtype=np.int32); trk = np.zeros(max_iter, dtype=np.int32)
        o.tr_rx, o.tr_ry, o.tr_ryaw, o.tr_nearest, o.tr_n_near = (trx.ctypes.data, try_.ctypes.data, tryaw.ctypes.data,
                                                                  trn.ctypes.data, trk.ctypes.data)
        o.tr_cap = int(max_iter)
    st = Stats()
    sob_idx = C.c_int64(0)
    if _rs_step > 0:
        rc = L.orc_plan_rrt_rs(C.byref(p), float(start[2]), float(goal[2]), float(curvature), float(goal_yaw_th),
                               float(goal_xy_th), float(_rs_step), obst.ctypes.data, len(obst), C.byref(rng), C.byref(o),
                               C.byref(st))
    elif _plain:
        rc = L.orc_plan_rrt_dubins(C.byref(p), float(start[2]), float(goal[2]), float(curvature), float(goal_yaw_th),
                                   float(goal_xy_th), obst.ctypes.data, len(obst), C.byref(rng), C.byref(o),
                                   C.byref(st), C.byref(sob_idx))
    else:
        rc = L.orc_plan_dubins(C.byref(p), float(start[2]), float(goal[2]), float(curvature), float(goal_yaw_th),
                               float(goal_xy_th), obst.ctypes.data, len(obst), C.byref(rng), C.byref(o), C.byref(st))
    if rc != 0:
        raise RuntimeError("orc_plan_dubins failed: %d" % rc)
    n = o.n
    res = dict(x=x[:n].copy(), y=y[:n].copy(), yaw=yaw[:n].copy(), cost=cost[:n].copy(), parent=parent[:n].copy(),
               poly_len=plen[:n].copy(), poly_x=ppx[:o.poly_n].copy(), poly_y=ppy[:o.poly_n].copy(),
               path=path[:o.path_n].copy() if o.path_n else None,
               path_yaw=path_yaw[:o.path_n].copy() if o.path_n else None,
               stats={k: getattr(st, k) for k, _ in Stats._fields_}, rng=rng, sobol_index=int(sob_idx.value))
    if trace:
        t = o.tr_n
        res.update(tr_rx=trx[:t].copy(), tr_ry=try_[:t].copy(), tr_ryaw=tryaw[:t].copy(), tr_nearest=trn[:t].copy(),
                   tr_n_near=trk[:t].copy())
    return res


class BOut(C.Structure):
    _fields_ = [("path_xy", C.c_void_p), ("path_cap", C.c_int32), ("path_n", C.c_int32),
                ("vertex_ids", C.c_void_p), ("g_scores", C.c_void_p), ("parent_ids", C.c_void_p),
                ("vcap", C.c_int32), ("nv", C.c_int32), ("n_edges", C.c_int32), ("n_samples", C.c_int32),
                ("tr_e0", C.c_void_p), ("tr_e1", C.c_void_p), ("tr_cap", C.c_int32), ("tr_n", C.c_int32),
                ("error", C.c_int32)]


def bitstar_rotation(start, goal):
    """cMin and C of rrt_08:189-202 (numpy SVD), exactly as the reference computes them."""
    import math
    c_min = math.hypot(start[0] - goal[0], start[1] - goal[1]) / 1.5
    a1 = np.array([[(goal[0] - start[0]) / c_min], [(goal[1] - start[1]) / c_min], [0]])
    id1_t = np.array([1.0, 0.0, 0.0]).reshape(1, 3)
    m = np.dot(a1, id1_t)
    u, s, vh = np.linalg.svd(m, True, True)
    c = np.dot(np.dot(u, np.diag([1.0, 1.0, np.linalg.det(u) * np.linalg.det(np.transpose(vh))])), vh)
    return c_min, c


def plan_bitstar(start, goal, obstacles, rand_area, max_iter=80, seed=None, rng=None, rot_c=None):
    """One BITStar(...).plan(animation=False) call of rrt_08 on the oracle."""
    L = lib()
    L.orc_plan_bitstar.restype = C.c_int
    L.orc_plan_bitstar.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    st = np.array([float(start[0]), float(start[1])]); gl = np.array([float(goal[0]), float(goal[1])])
    obst = np.ascontiguousarray(np.array(obstacles, dtype=np.float64).reshape(-1, 3))
    if rng is None:
        rng = mt_from_seed(seed)
    if rot_c is None:
        rot_c = bitstar_rotation(list(st), list(gl))[1]
    rc_ = np.ascontiguousarray(np.array(rot_c, dtype=np.float64).reshape(9))
    cap = 4096
    path = np.zeros((cap, 2)); vid = np.zeros(cap); g = np.zeros(cap); par = np.zeros(cap)
    tcap = 1 << 16
    t0 = np.zeros(tcap); t1 = np.zeros(tcap)
    o = BOut()
    o.path_xy, o.path_cap = path.ctypes.data, cap
    o.vertex_ids, o.g_scores, o.parent_ids, o.vcap = vid.ctypes.data, g.ctypes.data, par.ctypes.data, cap
    o.tr_e0, o.tr_e1, o.tr_cap = t0.ctypes.data, t1.ctypes.data, tcap
    rc = L.orc_plan_bitstar(st.ctypes.data, gl.ctypes.data, float(rand_area[0]), float(rand_area[1]), int(max_iter),
                            obst.ctypes.data, len(obst), rc_.ctypes.data, C.byref(rng), C.byref(o))
    if rc != 0:
        raise RuntimeError("orc_plan_bitstar failed: %d" % rc)
    return dict(path=path[:o.path_n].copy(), vertex_ids=vid[:o.nv].copy(), g_scores=g[:o.nv].copy(),
                parent_ids=par[:o.nv].copy(), n_edges=o.n_edges, n_samples=o.n_samples, error=o.error,
                tr_e0=t0[:min(o.tr_n, tcap)].copy(), tr_e1=t1[:min(o.tr_n, tcap)].copy(), rng=rng)


def path_smoothing(path, max_iter, obstacles, rng):
    """path_smoothing(path, max_iter, obstacle_list) of rrt_04:1447-1479 on the oracle; `rng` (MT state) advances."""
    L = lib()
    L.orc_path_smoothing.restype = C.c_int
    L.orc_path_smoothing.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p]
    pin = np.ascontiguousarray(np.array(path, dtype=np.float64).reshape(-1, 2))
    obst = np.ascontiguousarray(np.array(obstacles, dtype=np.float64).reshape(-1, 3))
    cap = len(pin) + int(max_iter) + 8
    out = np.zeros((cap, 2))
    n = C.c_int(0)
    rc = L.orc_path_smoothing(pin.ctypes.data, len(pin), int(max_iter), obst.ctypes.data, len(obst), C.byref(rng),
                              out.ctypes.data, cap, C.byref(n))
    if rc != 0:
        raise RuntimeError("orc_path_smoothing failed: %d" % rc)
    return out[:n.value].copy()


def reeds_shepp(sx, sy, syaw, gx, gy, gyaw, maxc, step_size=0.2):
    """reeds_shepp_path_planning (rrt_06:1426-1441) on the oracle -> (px, py, pyaw, mode, lengths), all None where the
    reference returns None; raises ZeroDivisionError / ValueError where the reference does."""
    L = lib()
    L.orc_reeds_shepp.restype = C.c_int
    L.orc_reeds_shepp.argtypes = [C.c_double] * 8 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p,
                                                     C.c_void_p]
    cap = 8192
    px = np.zeros(cap); py = np.zeros(cap); pyaw = np.zeros(cap); ln = np.zeros(5)
    mode = C.create_string_buffer(8)
    nl = C.c_int(0)
    n = L.orc_reeds_shepp(float(sx), float(sy), float(syaw), float(gx), float(gy), float(gyaw), float(maxc),
                          float(step_size), px.ctypes.data, py.ctypes.data, pyaw.ctypes.data, cap, ln.ctypes.data, mode,
                          C.byref(nl))
    if n == -3:
        raise ZeroDivisionError("float division by zero")
    if n == -4:
        raise ValueError("math domain error")
    if n < 0:
        raise RuntimeError("orc_reeds_shepp: capacity")
    if n == 0:
        return None, None, None, None, None
    return px[:n].copy(), py[:n].copy(), pyaw[:n].copy(), mode.value.decode(), ln[:nl.value].copy()
