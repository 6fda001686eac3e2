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
