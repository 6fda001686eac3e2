"""CPU: the oracle restatement reproduces every golden vector generated from the reference itself."""
import numpy as np
import pytest

import util


@pytest.mark.parametrize("path", util.golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_matches_reference_golden(path):
    g = util.load_golden(path)
    kw = util.kwargs_from_golden(g)
    r = util.run_oracle(kw, int(g["seed"]), exact_pow=True, trace=True)
    util.assert_tree_equal((r["x"], r["y"], r["cost"], r["parent"]),
                           (g["x"], g["y"], g["cost"] if kw["algo"] == "rrt_star" else None, g["parent"]), g["name"])
    if len(g["path"]) == 0:
        assert r["path"] is None
    else:
        assert r["path"] is not None and np.array_equal(r["path"], g["path"])
    assert r["stats"]["edges_ref"] == int(g["ref_edges"])          # same number of check_collision calls
    assert r["rng"].pos == int(g["rng_pos_after"]) and r["rng"].mt[0] == int(g["rng_word0_after"])
    n = len(g["tr_nearest"])
    assert np.array_equal(r["tr_nearest"][:n], g["tr_nearest"])
    assert np.array_equal(r["tr_rnd_x"][:n], g["tr_rnd_x"]) and np.array_equal(r["tr_rnd_y"][:n], g["tr_rnd_y"])
    if kw["sobol"]:
        assert r["stats"]["sobol_index"] == int(g["sobol_index_after"])


@pytest.mark.parametrize("it", [1000, 4000])
def test_guarded_square_equals_exact(it):
    """The guarded x*x form used above oracle-feasible sizes takes the same decisions as always-pow."""
    kw = util.c2_kwargs(it)
    a = util.run_oracle(kw, 1, exact_pow=True)
    b = util.run_oracle(kw, 1, exact_pow=False)
    util.assert_tree_equal((a["x"], a["y"], a["cost"], a["parent"]), (b["x"], b["y"], b["cost"], b["parent"]))
    assert b["stats"]["pow_slow"] > 0


def test_spot_values_survey_section10():
    g = util.load_golden(util.GOLDEN + "/rrt04_c2_s1_it8000.npz")
    assert len(g["x"]) == 7685 and int(g["ref_edges"]) == 286894
    p = g["path"]
    import math
    le = sum(math.hypot(p[i + 1][0] - p[i][0], p[i + 1][1] - p[i][1]) for i in range(len(p) - 1))
    assert abs(le - 144.382254) < 1e-6


@pytest.mark.parametrize("path", util.golden_files("rrt07"), ids=lambda p: p.split("/")[-1][:-4])
def test_informed_oracle_matches_reference_golden(path):
    """rrt_07 Informed RRT*: tree, best path, c_best, RNG state and per-iteration samples (incl. the ellipsoidal
    informed samples, i.e. the numpy dot forms) equal the reference's, bit for bit."""
    import oracle
    g = util.load_golden(path)
    kw = util.informed_kwargs_from_golden(g)
    assert np.array_equal(oracle.rotation_to_world(kw["start"], kw["goal"]), g["rot_c"])
    r = oracle.plan_informed(seed=int(g["seed"]), trace=True, **kw)
    util.assert_tree_equal((r["x"], r["y"], r["cost"], r["parent"]), (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    if len(g["path"]) == 0:
        assert r["path"] is None and r["c_best"] == float("inf")
    else:
        assert np.array_equal(r["path"], g["path"]) and r["c_best"] == float(g["path_len"])
    assert r["rng"].pos == int(g["rng_pos_after"]) and r["rng"].mt[0] == int(g["rng_word0_after"])
    n = len(g["tr_nearest"])
    assert np.array_equal(r["tr_nearest"][:n], g["tr_nearest"])
    assert np.array_equal(r["tr_rnd_x"][:n], g["tr_rnd_x"]) and np.array_equal(r["tr_rnd_y"][:n], g["tr_rnd_y"])
    if kw["sobol"]:
        assert r["stats"]["sobol_index"] == int(g["sobol_index_after"])


def test_informed_spot_value_survey_section10():
    g = util.load_golden(util.GOLDEN + "/rrt07_drv_mt_s42_it2000.npz")
    assert len(g["x"]) == 1397 and len(g["path"]) == 13 and float(g["path_len"]) == 17.33495114327498


def test_dubins_known_answers():
    """plan_dubins_path (rrt_05:1021-1109) on 400 random pose pairs: polyline, word, lengths and final yaw equal the
    reference's bit for bit (pins the scipy rotation form, the numpy matmul forms and numpy's `%`)."""
    import oracle
    k = np.load(util.GOLDEN + "/dubins_kat.npz")
    off = 0
    for i in range(len(k["n"])):
        n = int(k["n"][i])
        px, py, pyaw, mode, ln = oracle.dubins(*k["inp"][i])
        assert len(px) == n and mode == str(k["mode"][i])
        assert np.array_equal(px, k["poly_x"][off:off + n]) and np.array_equal(py, k["poly_y"][off:off + n])
        assert np.array_equal(ln, k["lengths"][i]) and pyaw[-1] == k["end"][i][2]
        off += n
    # SURVEY.md section 10 KAT: (1,1,45deg) -> (-3,-3,-45deg), curvature 1
    import math
    px, py, pyaw, mode, ln = oracle.dubins(1.0, 1.0, math.radians(45.0), -3.0, -3.0, math.radians(-45.0))
    assert mode == "LSL" and len(px) == 94
    assert list(ln) == [3.3531176436132273, 4.76301285963152, 1.3592713367714622]
    assert (px[-1], py[-1], pyaw[-1]) == (-3.000000000000001, -2.9999999999999987, -0.7853981633974492)


@pytest.mark.parametrize("path", util.golden_files("rrt05"), ids=lambda p: p.split("/")[-1][:-4])
def test_dubins_rrt_star_oracle_matches_reference_golden(path):
    """rrt_05 RRT*-Dubins: poses, costs, parents, every stored edge polyline, final course and RNG state."""
    import oracle
    g = util.load_golden(path)
    r = oracle.plan_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], int(g["max_iter"]), seed=int(g["seed"]),
                           trace=True, search_until_max_iter=bool(int(g.get("search_until_max_iter", 1))))
    util.assert_tree_equal((r["x"], r["y"], r["cost"], r["parent"]), (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    assert np.array_equal(r["yaw"], g["yaw"])
    assert np.array_equal(r["poly_len"], g["poly_len"]) and np.array_equal(r["poly_x"], g["poly_x"]) \
        and np.array_equal(r["poly_y"], g["poly_y"])
    if len(g["path"]) == 0:
        assert r["path"] is None
    else:
        assert r["path"] is not None and np.array_equal(r["path"], g["path"])
    assert r["rng"].pos == int(g["rng_pos_after"]) and r["rng"].mt[0] == int(g["rng_word0_after"])
    n = len(g["tr_nearest"])
    assert np.array_equal(r["tr_nearest"][:n], g["tr_nearest"]) and np.array_equal(r["tr_ryaw"][:n], g["tr_ryaw"])


@pytest.mark.parametrize("path", util.golden_files("rrt06"), ids=lambda p: p.split("/")[-1][:-4])
def test_reeds_shepp_rrt_star_oracle_matches_reference_golden(path):
    """rrt_06 RRT*-Reeds-Shepp (oracle only so far): poses, costs, parents incl. the try_goal_path
    nodes, every stored edge polyline, the three-column final course and the RNG state."""
    import oracle
    g = util.load_golden(path)
    r = oracle.plan_rrt_rs(g["start"], g["goal"], g["obstacles"], g["rand_area"], int(g["max_iter"]), seed=int(g["seed"]),
                           curvature=float(g["curvature"]), robot_radius=float(g["robot_radius"]),
                           expand_dis=float(g["expand_dis"]), connect_circle_dist=float(g["connect_circle_dist"]),
                           step_size=float(g["step_size"]), trace=True,
                           search_until_max_iter=bool(int(g["search_until_max_iter"])))
    util.assert_tree_equal((r["x"], r["y"], r["cost"], r["parent"]), (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    assert np.array_equal(r["yaw"], g["yaw"])
    assert np.array_equal(r["poly_len"], g["poly_len"]) and np.array_equal(r["poly_x"], g["poly_x"]) \
        and np.array_equal(r["poly_y"], g["poly_y"])
    if len(g["path"]) == 0:
        assert r["path"] is None
    else:
        assert r["path"] is not None and np.array_equal(r["path"], g["path"][:, :2])
        assert np.array_equal(r["path_yaw"], g["path"][:, 2])
    assert r["rng"].pos == int(g["rng_pos_after"]) and r["rng"].mt[0] == int(g["rng_word0_after"])
    n = len(g["tr_nearest"])
    assert np.array_equal(r["tr_nearest"][:n], g["tr_nearest"]) and np.array_equal(r["tr_ryaw"][:n], g["tr_ryaw"])


def test_lazy_candidate_order_builds_the_same_tree_dubins():
    """The same argument for rrt_05 (the kernel's opt-in RRTX_DUBINS_LAZY=1): oracle in the lazy order vs the goldens."""
    import oracle
    oracle.set_lazy_order(True)
    try:
        for path in util.golden_files("rrt05_drv"):
            g = util.load_golden(path)
            r = oracle.plan_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], int(g["max_iter"]),
                                   seed=int(g["seed"]))
            util.assert_tree_equal((r["x"], r["y"], r["cost"], r["parent"]), (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
            assert np.array_equal(r["poly_x"], g["poly_x"]) and r["rng"].pos == int(g["rng_pos_after"])
    finally:
        oracle.set_lazy_order(False)


def test_lazy_candidate_order_builds_the_same_tree():
    """The rrt_06 kernel steers choose_parent / rewire candidates lazily (rrt_rs.hip.h); the argument that this cannot
    change the tree, checked on the CPU: the oracle in that order vs the oracle in the reference's order (every candidate
    steered), 40 seeds of the driver scene and 12 of a second scene; it must also steer far fewer edges."""
    import oracle
    g = util.load_golden(util.GOLDEN + "/rrt06_drv_s42_it200.npz")
    cases = [(s, dict(curvature=2.0, robot_radius=0.6, step_size=0.1), [10.0, 9.0, 0.0], 400) for s in range(200, 240)]
    cases += [(s, dict(curvature=1.0, robot_radius=0.0, step_size=0.2), [10.0, 9.0, 1.2], 600) for s in range(300, 312)]
    e_eager = e_lazy = 0
    for seed, kw, goal, it in cases:
        ref = oracle.plan_rrt_rs(g["start"], goal, g["obstacles"], g["rand_area"], it, seed=seed, **kw)
        oracle.set_lazy_order(True)
        try:
            lz = oracle.plan_rrt_rs(g["start"], goal, g["obstacles"], g["rand_area"], it, seed=seed, **kw)
        finally:
            oracle.set_lazy_order(False)
        util.assert_tree_equal((lz["x"], lz["y"], lz["cost"], lz["parent"]), (ref["x"], ref["y"], ref["cost"], ref["parent"]),
                               "seed %d" % seed)
        assert np.array_equal(lz["yaw"], ref["yaw"]) and np.array_equal(lz["poly_x"], ref["poly_x"])
        assert (lz["path"] is None) == (ref["path"] is None)
        if ref["path"] is not None:
            assert np.array_equal(lz["path"], ref["path"]) and np.array_equal(lz["path_yaw"], ref["path_yaw"])
        assert lz["rng"].pos == ref["rng"].pos
        e_eager += ref["stats"]["edges_unique"]
        e_lazy += lz["stats"]["edges_unique"]
    assert e_lazy * 4 < e_eager


@pytest.mark.parametrize("path", util.golden_files("rrt03"), ids=lambda p: p.split("/")[-1][:-4])
def test_rrt_dubins_oracle_matches_reference_golden(path):
    """rrt_03 RRT with Dubins steer (pseudo-random and 3-D Sobol sampler): poses, Dubins-length costs, parents, every
    stored edge polyline, final course, RNG state and Sobol index."""
    import oracle
    g = util.load_golden(path)
    r = oracle.plan_rrt_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], int(g["max_iter"]),
                               seed=int(g["seed"]), robot_radius=float(g["robot_radius"]),
                               goal_sample_rate=int(g["goal_sample_rate"]), sobol=bool(int(g["sobol"])), trace=True,
                               search_until_max_iter=bool(int(g.get("search_until_max_iter", 1))))
    util.assert_tree_equal((r["x"], r["y"], r["cost"], r["parent"]), (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    assert np.array_equal(r["yaw"], g["yaw"])
    assert np.array_equal(r["poly_len"], g["poly_len"]) and np.array_equal(r["poly_x"], g["poly_x"]) \
        and np.array_equal(r["poly_y"], g["poly_y"])
    if len(g["path"]) == 0:
        assert r["path"] is None
    else:
        assert r["path"] is not None and np.array_equal(r["path"], g["path"])
    assert r["rng"].pos == int(g["rng_pos_after"]) and r["rng"].mt[0] == int(g["rng_word0_after"])
    if int(g["sobol"]):
        assert r["sobol_index"] == int(g["sobol_index_after"])
    n = len(g["tr_nearest"])
    assert np.array_equal(r["tr_nearest"][:n], g["tr_nearest"]) and np.array_equal(r["tr_ryaw"][:n], g["tr_ryaw"])
    assert np.array_equal(r["tr_rx"][:n], g["tr_rx"])


@pytest.mark.parametrize("path", util.golden_files("rrt08"), ids=lambda p: p.split("/")[-1][:-4])
def test_bitstar_oracle_matches_reference_golden(path):
    """rrt_08 BIT*: vertex insertion order, g-scores, parents, returned path, tree edge / sample counts, the whole
    sequence of edges popped from the edge queue, and the RNG state equal the reference's."""
    import oracle
    g = util.load_golden(path)
    c_min, c = oracle.bitstar_rotation(list(g["start"]), list(g["goal"]))
    assert np.array_equal(c, g["rot_c"]) and c_min == float(g["c_min"])
    r = oracle.plan_bitstar(g["start"], g["goal"], g["obstacles"], g["rand_area"], int(g["max_iter"]), seed=int(g["seed"]))
    assert r["error"] == (1 if str(g["error"]) else 0)
    assert np.array_equal(r["vertex_ids"], g["vertex_ids"]) and np.array_equal(r["g_scores"], g["g_scores"])
    assert np.array_equal(r["parent_ids"], g["parent_ids"])
    assert np.array_equal(r["path"], g["path"])
    assert r["n_edges"] == int(g["n_edges"]) and r["n_samples"] == int(g["n_samples"])
    assert np.array_equal(r["tr_e0"], g["tr_e0"]) and np.array_equal(r["tr_e1"], g["tr_e1"])
    assert r["rng"].pos == int(g["rng_pos_after"]) and r["rng"].mt[0] == int(g["rng_word0_after"])


def test_bitstar_spot_value_survey_section10():
    g = util.load_golden(util.GOLDEN + "/rrt08_s42_it80.npz")
    p = g["path"]
    assert len(p) == 9 and len(g["vertex_ids"]) == 80 and int(g["n_edges"]) == 79
    assert np.allclose(p[:4], [[-1.0, 0.0], [-0.64, 1.16], [-0.28, 2.73], [-0.92, 4.49]], atol=1e-12)


@pytest.mark.parametrize("path", util.golden_files("smooth"), ids=lambda p: p.split("/")[-1][:-4])
def test_path_smoothing_oracle_matches_reference_golden(path):
    """path_smoothing (rrt_04:1447-1479) after planning, on the stream planning() left: smoothed polyline and RNG state."""
    import oracle
    g = util.load_golden(path)
    rng = oracle.MT()
    for i in range(624):
        rng.mt[i] = int(g["rng_mt_before"][i])
    rng.pos = int(g["rng_pos_before"])
    sm = oracle.path_smoothing(g["path_in"], int(g["max_iter"]), g["obstacles"], rng)
    assert np.array_equal(sm, g["smoothed"])
    assert rng.pos == int(g["rng_pos_after"]) and rng.mt[0] == int(g["rng_word0_after"])


def test_reeds_shepp_oracle_matches_reference_kat():
    """Groundwork for rrt_06: the oracle's Reeds-Shepp solver against 600 known-answer vectors of the
    reference's reeds_shepp_path_planning (word, segment lengths, every path point and yaw; None and raising cases)."""
    import oracle
    g = np.load(util.GOLDEN + "/rs_kat.npz")
    off = 0
    for k in range(len(g["inp"])):
        a = [float(v) for v in g["inp"][k]]
        n = int(g["n"][k])
        if n < 0:
            with pytest.raises((ZeroDivisionError, ValueError)) as ei:
                oracle.reeds_shepp(*a)
            assert ei.type.__name__ == str(g["mode"][k]), k
            continue
        px, py, pyaw, mode, ln = oracle.reeds_shepp(*a)
        if n == 0:
            assert px is None, k
            continue
        assert px is not None and len(px) == n, (k, n, None if px is None else len(px))
        assert mode == str(g["mode"][k]), k
        assert np.array_equal(ln, g["lengths"][k][:int(g["n_len"][k])]), k
        assert np.array_equal(px, g["poly_x"][off:off + n]) and np.array_equal(py, g["poly_y"][off:off + n]), k
        assert np.array_equal(pyaw, g["poly_yaw"][off:off + n]), k
        off += n


def test_oracle_diagnostic_counters_are_exported():
    """The counters the device-side shortcuts are argued with: moved / revisited nodes of rrt_04's
    rewire and late-qualifying rewire candidates of rrt_05 -- present, and moving in the expected direction."""
    import ctypes as C
    import oracle
    L = oracle.lib()
    m0, r0, m1, r1 = C.c_long(), C.c_long(), C.c_long(), C.c_long()
    L.orc_moved_counters(C.byref(m0), C.byref(r0))
    kw = dict(util.C2)
    kw.update(start=[0, 0], goal=[6, 8], rand_area=[-2, 12], obstacles=[(3, 3, 1)], expand_dis=3.0, path_resolution=0.05,
              goal_sample_rate=60, max_iter=400)
    util.run_oracle(kw, 5, exact_pow=True)
    L.orc_moved_counters(C.byref(m1), C.byref(r1))
    assert m1.value - m0.value == 17 and r1.value == r0.value     # tools/find_moved_node.py: 17 moved nodes, no second visit
    q, w = C.c_long(), C.c_long()
    L.orc_late_counters(C.byref(q), C.byref(w))
    assert q.value >= w.value >= 0
