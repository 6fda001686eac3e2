"""GPU (-m gpu): the HIP path, called through the C ABI, against the golden vectors generated from the
reference and against the CPU oracle on the same seeded inputs.  Integer arrays bit-exact; doubles are
compared bit-exact as well (stronger than the 1e-6 the contract asks for)."""
import math

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def test_device_arithmetic_replicas(gpu):
    """hypot / **2 / sin / cos / atan2 / steer / sqrt / division on the device == CPython+glibc on the host."""
    import oracle
    import rrt_amd
    L = oracle.lib()
    rng = np.random.default_rng(7)
    n = 200000
    a = (rng.random(n) * 2 - 1) * 120
    b = (rng.random(n) * 2 - 1) * 120
    a[::7] *= 1e-3
    b[::11] *= 1e-6
    a[::1013] = 0.0
    th = (rng.random(n) * 2 - 1) * 3.2
    sel = rrt_amd._abi.selftest_math
    got = sel(0, a, b)
    assert all(got[i] == math.hypot(a[i], b[i]) for i in range(n)), "hypot"
    got = sel(1, a, b)
    assert all(got[i] == (a[i] ** 2) for i in range(n)), "**2"
    got = sel(2, th, b)
    assert all(got[i] == math.sin(th[i]) for i in range(n)), "sin"
    got = sel(3, th, b)
    assert all(got[i] == math.cos(th[i]) for i in range(n)), "cos"
    got = sel(4, a, b)
    assert all(got[i] == math.atan2(a[i], b[i]) for i in range(n)), "atan2"
    got = sel(6, np.abs(a), b)
    assert all(got[i] == math.sqrt(abs(a[i])) for i in range(n)), "sqrt"
    bb = np.where(b == 0, 1.0, b)
    got = sel(7, a, bb)
    assert np.array_equal(got, a / bb), "division"
    u = rng.random(n) * 2 - 1
    u[::17] *= 1e-4
    u[::19] = np.sign(u[::19]) * (1 - np.abs(u[::19]) * 1e-6)
    got = sel(8, u, b)
    assert all(got[i] == math.acos(u[i]) for i in range(n)), "acos"
    got = sel(9, u, b)
    assert all(got[i] == math.asin(u[i]) for i in range(n)), "asin"
    # Reeds-Shepp steer core (rpp_rs.h, the steer of the rrt_06 kernel) on the device vs the oracle's restatement
    m = 3000
    gx = (rng.random(m) * 2 - 1) * 8
    gy = (rng.random(m) * 2 - 1) * 8
    got = sel(10, gx, gy)
    for i in range(m):
        try:
            px, py, pyaw, mode, ln = oracle.reeds_shepp(0.0, 0.0, 0.0, gx[i], gy[i], gx[i] + gy[i], 1.0, 0.2)
        except (ZeroDivisionError, ValueError):
            continue
        if px is None or len(px) > 256:
            continue
        k = len(px)
        assert got[i] == px[k - 1] + py[k // 2] + pyaw[k - 1] + ln[0], ("reeds-shepp", i)
    # steer end point (constructed near-ties included): (0,0) -> (a,b) scaled to length ~2.0
    ang = th
    tx = 2.0 * np.cos(ang)
    ty = 2.0 * np.sin(ang)
    got = sel(5, tx, ty)
    import ctypes as C
    end = (C.c_double * 2)()
    sn = C.c_int()
    px = (C.c_double * 64)()
    py = (C.c_double * 64)()
    L.orc_steer_polyline.argtypes = [C.c_double] * 6 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    for i in range(0, n, 10):
        L.orc_steer_polyline(0.0, 0.0, tx[i], ty[i], float("inf"), 0.25, px, py, 64, end, C.byref(sn))
        assert got[i] == end[0], i


def _check_against_golden(g, out, i=0):
    kw = util.kwargs_from_golden(g)
    util.assert_tree_equal(out["trees"][i], (g["x"], g["y"], g["cost"] if kw["algo"] == "rrt_star" else None,
                                             g["parent"]), g["name"])
    p = out["paths"][i]
    if len(g["path"]) == 0:
        assert p is None
    else:
        assert p is not None and np.array_equal(p, g["path"])
    st = out["rng"][i]
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])


def test_runtime_selfcheck_of_the_arithmetic_contract(gpu):
    """rrtx_selfcheck (native: device replicas vs the HOST's libm) + the Python side (math.hypot, float ** 2 vs this
    interpreter): on the image the goldens come from every count is zero and no RrtxParityWarning is raised; the first
    Handle of a process runs it by itself."""
    import warnings
    import rrt_amd
    A = rrt_amd._abi
    A._selfchecked.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error", A.RrtxParityWarning)
        res = A.selfcheck(0, n=20000)
        assert set(res) == set(A.SELFCHECK_FUNCS) and all(v == 0 for v in res.values()), res
        assert A.selfcheck(0) is res            # cached per device
        A._selfchecked.clear()
        h = A.Handle(A.ALGO_RRT, [0, 0], [6, 10], [-2, 15], 1.0, 0.1, 5, 10)   # the automatic call
        h.close()
        assert 0 in A._selfchecked and not any(A._selfchecked[0].values())


@pytest.mark.parametrize("path", util.golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_gpu_matches_reference_golden(gpu, path):
    g = util.load_golden(path)
    kw = util.kwargs_from_golden(g)
    out = util.run_gpu_batch(kw, [int(g["seed"])], trace_instance=0)
    d = util.first_trace_divergence(out["trace"], g["tr_rnd_x"], g["tr_rnd_y"], g["tr_nearest"])
    assert d is None, "first divergent iteration %d" % d
    _check_against_golden(g, out)
    assert out["stats"]["edges_ref"] == int(g["ref_edges"])
    if kw["sobol"]:
        assert out["sobol_index"][0] == int(g["sobol_index_after"])


def test_gpu_batch_equals_oracle_per_seed(gpu):
    """Many instances in one launch: instance i == oracle(seed i), independent of its neighbours."""
    kw = util.c2_kwargs(1500)
    seeds = list(range(1, 41))
    out = util.run_gpu_batch(kw, seeds)
    pc, nn, st = out["results"]
    for i, s in enumerate(seeds):
        r = util.run_oracle(kw, s, exact_pow=True)
        util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "seed %d" % s)
        if r["path"] is None:
            assert out["paths"][i] is None and math.isinf(pc[i])
        else:
            assert np.array_equal(out["paths"][i], r["path"])
            le = 0.0
            for j in range(len(r["path"]) - 1):
                le += math.hypot(r["path"][j + 1][0] - r["path"][j][0], r["path"][j + 1][1] - r["path"][j][1])
            assert pc[i] == le
        assert nn[i] == len(r["x"])
    agg = out["stats"]
    assert agg["total_nodes"] == int(nn.sum())


def test_gpu_stats_match_oracle_counters(gpu):
    kw = util.c2_kwargs(3000)
    out = util.run_gpu_batch(kw, [5])
    r = util.run_oracle(kw, 5, exact_pow=True)
    s, o = out["stats"], r["stats"]
    for k in ("edges_ref", "edges_unique", "near_hits", "near_unique", "rewires", "propagated", "iterations"):
        assert s[k] == o[k], k


def test_gpu_c2_20k_nodes_equals_oracle(gpu):
    """Well past the reference-feasible size: GPU vs the (golden-pinned) oracle, 20 000 iterations."""
    kw = util.c2_kwargs(20000)
    out = util.run_gpu_batch(kw, [1, 2])
    for i, s in enumerate([1, 2]):
        r = util.run_oracle(kw, s, exact_pow=False)
        util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "seed %d" % s)
        assert np.array_equal(out["paths"][i], r["path"])


def test_gpu_c2_full_size_equals_oracle(gpu, monkeypatch):
    """BASELINE.json's full size (C2: 105 000 iterations, ~101 k nodes) against the golden-pinned oracle, bit for bit --
    tree, path, counters -- on the kernel shape bench.py times: the 64-thread workgroup `rppk2t::rrt_star_kernel_v2<true>`
    with the 16-bit first stage (a 1-instance handle would pick the 256-thread shape by itself; RRTX_TPB pins it).
    ~2 min of oracle time on one host core."""
    monkeypatch.setenv("RRTX_TPB", "64")
    kw = util.c2_kwargs(105000)
    out = util.run_gpu_batch(kw, [1])
    r = util.run_oracle(kw, 1, exact_pow=False)
    util.assert_tree_equal(out["trees"][0], (r["x"], r["y"], r["cost"], r["parent"]), "seed 1, 105k")
    assert np.array_equal(out["paths"][0], r["path"])
    for k in ("edges_ref", "edges_unique", "near_hits", "near_unique", "rewires", "propagated", "iterations"):
        assert out["stats"][k] == r["stats"][k], k
    assert out["stats"]["q16_fallbacks"] > 0      # the 16-bit stage ran (and handed some queries down)
    assert out["stats"]["passes_shared"] > 55000  # ... and more than half of the iterations rode on an earlier pass


def _orc_c2(a):
    kw, sd = a
    r = util.run_oracle(kw, sd, exact_pow=False)
    return r["x"], r["y"], r["cost"], r["parent"], r["path"]


def test_gpu_c2_bench_shape_batch_sampled_against_oracle(gpu):
    """The bench's occupancy regime: more than 2 560 instances in one handle, so the library selects the 64-thread
    shape by itself (16 workgroups per CU, every instance contending for the CU's memory pipeline with 15 others).
    3 072 instances x 3 000 iterations; 32 of them, spread over the batch, against the oracle bit for bit."""
    import concurrent.futures as cf
    kw = util.c2_kwargs(3000)
    import rrt_amd
    A = rrt_amd._abi
    B = 3072
    seeds = list(range(1, B + 1))
    h = A.Handle(A.ALGO_RRT_STAR, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], kw["path_resolution"],
                 kw["goal_sample_rate"], kw["max_iter"], robot_radius=0.0, connect_circle_dist=50.0,
                 search_until_max_iter=True, n_instances=B)
    try:
        h.set_obstacles(kw["obstacles"])
        h.seed_instances(seeds)
        assert h.plan() == 0
        pick = [int(v) for v in np.linspace(0, B - 1, 32)]
        with cf.ProcessPoolExecutor(max_workers=8) as ex:
            orc = list(ex.map(_orc_c2, [(kw, seeds[i]) for i in pick]))
        for i, (ox, oy, oc, op, opath) in zip(pick, orc):
            util.assert_tree_equal(h.get_tree(i), (ox, oy, oc, op), "instance %d" % i)
            p = h.get_path(i)
            assert (p is None) == (opath is None) and (p is None or np.array_equal(p, opath))
    finally:
        h.close()


def test_gpu_near_set_overflow_is_replanned_on_a_larger_shape(gpu, monkeypatch):
    """A near set that outgrows the 64-thread shape's 44 LDS candidate slots (the reference driver's 17 x 17 scene with
    expand_dis 3 reaches ~50 neighbours within a few hundred nodes): the affected instances are planned again on the
    next larger shape instead of failing the call (round-1 ADVICE).  RRTX_TPB pins the first attempt to the small
    shape (the library's own estimate would not pick it for this scene); results equal the oracle's."""
    g = util.load_golden(util.GOLDEN + "/rrt04_drv_mt_s1234.npz") if False else None
    kw = dict(util.C2)
    kw.update(start=[0, 0], goal=[6, 10], rand_area=[-2, 15], expand_dis=3.0, path_resolution=0.5, max_iter=700,
              obstacles=[(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)], robot_radius=0.8)
    monkeypatch.setenv("RRTX_TPB", "64")
    seeds = list(range(1, 9))
    out = util.run_gpu_batch(kw, seeds)     # plan(strict=True): raises if any instance is left with ST_OVERFLOW
    assert out["stats"]["near_unique_max"] > 44
    for i, s in enumerate(seeds):
        r = util.run_oracle(kw, s, exact_pow=True)
        util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "seed %d" % s)
    monkeypatch.setenv("RRTX_NO_RETRY", "1")
    import rrt_amd
    with pytest.raises(rrt_amd._abi.RrtxError):
        util.run_gpu_batch(kw, seeds)


def test_gpu_c2_full_size_kernel_variants_agree(gpu, monkeypatch):
    """Full size, every code path of the RRT* iteration kernel: three workgroup shapes x {f32-mirror pass, f64 pass}
    must produce identical trees (different reductions, filters and fallbacks, one reference semantics)."""
    kw = util.c2_kwargs(105000)
    seeds = [2, 3, 4]
    base = None
    for tpb, f32, spec2 in (("256", "0", "3"), ("256", "1", "3"), ("128", "1", "3"), ("64", "1", "3"), ("64", "1", "1"),
                            ("64", "1", "0"), ("64", "0", "3")):
        monkeypatch.setenv("RRTX_TPB", tpb)
        monkeypatch.setenv("RRTX_F32", f32)
        monkeypatch.setenv("RRTX_SPEC2", spec2)   # 64-thread shape: a streaming pass serves up to 1 + spec2 iterations
        out = util.run_gpu_batch(kw, seeds)
        # most iterations of a dense tree ride on an earlier pass -- and only in that configuration
        rides = out["stats"]["passes_shared"]
        if (tpb, f32) == ("64", "1") and spec2 != "0":
            assert rides > (0.55 if spec2 == "3" else 0.3) * out["stats"]["iterations"], (spec2, rides)
        else:
            assert rides == 0
        sig = [tuple(np.ascontiguousarray(a).tobytes() for a in t) for t in out["trees"]]
        paths = [None if p is None else np.asarray(p).tobytes() for p in out["paths"]]
        if base is None:
            base = (sig, paths, out["stats"]["rewires"], out["stats"]["propagated"])
        else:
            assert sig == base[0] and paths == base[1], (tpb, f32, spec2)
            assert (out["stats"]["rewires"], out["stats"]["propagated"]) == base[2:], (tpb, f32, spec2)


def test_size_independent_invariants(gpu):
    """Properties that hold at any size: tree consistency after planning."""
    kw = util.c2_kwargs(6000)
    out = util.run_gpu_batch(kw, [11])
    x, y, cost, parent = out["trees"][0]
    assert parent[0] == -1 and (parent[1:] >= 0).all() and (parent[1:] < len(x)).all()
    for i in range(1, len(x)):
        p = parent[i]
        assert cost[i] == cost[p] + math.hypot(x[i] - x[p], y[i] - y[p])   # bitwise, rrt_04:1379-1384
    # acyclic: every node reaches the root
    depth = np.zeros(len(x), dtype=np.int64)
    for i in range(1, len(x)):
        j, d = i, 0
        while j != 0:
            j = parent[j]
            d += 1
            assert d <= len(x)
        depth[i] = d


def test_host_classes_drop_in(gpu):
    """The reference's driver usage (rrt_04:1532-1580) against the mirror classes."""
    import random
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt04_drv_mt_s1234.npz")
    kw = util.kwargs_from_golden(g)
    random.seed(1234)
    rrt = rrt_amd.RRTStar(start=kw["start"], goal=kw["goal"], obstacle_list=kw["obstacles"], rand_area=kw["rand_area"],
                          expand_dis=kw["expand_dis"], path_resolution=kw["path_resolution"],
                          goal_sample_rate=kw["goal_sample_rate"], max_iter=kw["max_iter"], play_area=kw["play_area"],
                          robot_radius=kw["robot_radius"], sobol_sampler=False, connect_circle_dist=50.0,
                          search_until_max_iter=True)
    path = rrt.planning(animation=False)
    assert len(rrt.node_list) == 150 and len(path) == 24
    assert rrt_amd.get_path_length(path) == 21.309028648444052
    assert path[1] == [5.780590352994788, 10.488215966400231]
    assert random.getstate()[1][624] == int(g["rng_pos_after"])           # global stream advanced as the reference would
    nd = rrt.node_list[1]
    assert nd.parent is rrt.node_list[int(g["parent"][1])] and len(nd.path_x) >= 2
    g1 = util.load_golden(util.GOLDEN + "/rrt01_drv_s42.npz")
    random.seed(42)
    r1 = rrt_amd.RRT(start=kw["start"], goal=kw["goal"], obstacle_list=kw["obstacles"], rand_area=kw["rand_area"],
                     expand_dis=1.0, path_resolution=0.1, goal_sample_rate=5, max_iter=500, play_area=None,
                     robot_radius=0.6)
    p1 = r1.planning(animation=False)
    assert len(r1.node_list) == 170 and len(p1) == 28 and np.array_equal(np.array(p1), g1["path"])


@pytest.mark.parametrize("path", util.golden_files("rrt07"), ids=lambda p: p.split("/")[-1][:-4])
def test_gpu_informed_matches_reference_golden(gpu, path):
    """rrt_07 Informed RRT* on the GPU vs the reference goldens (ints exact; doubles compared bit-exact, which holds
    on this image because the numpy dot forms are restated as measured -- contract tolerance for floats is 1e-6)."""
    g = util.load_golden(path)
    kw = util.informed_kwargs_from_golden(g)
    out = util.run_gpu_informed(kw, [int(g["seed"])], trace_instance=0)
    d = util.first_trace_divergence(out["trace"], g["tr_rnd_x"], g["tr_rnd_y"], g["tr_nearest"])
    assert d is None, "first divergent iteration %d" % d
    x, y, cost, parent = out["trees"][0]
    assert len(x) == len(g["x"]) and np.array_equal(parent, g["parent"])
    assert np.allclose(x, g["x"], rtol=0, atol=1e-6) and np.allclose(cost, g["cost"], rtol=0, atol=1e-6)
    util.assert_tree_equal(out["trees"][0], (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    p = out["paths"][0]
    if len(g["path"]) == 0:
        assert p is None
    else:
        assert p is not None and np.array_equal(p, g["path"])
        assert out["results"][0][0] == float(g["path_len"])
    st = out["rng"][0]
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])


def test_gpu_informed_batch_equals_oracle(gpu):
    import oracle
    g = util.load_golden(util.GOLDEN + "/rrt07_c3_sobol_s1_it3000.npz")
    kw = util.informed_kwargs_from_golden(g)
    kw["max_iter"] = 1500
    seeds = list(range(1, 13))
    out = util.run_gpu_informed(kw, seeds)
    for i, s in enumerate(seeds):
        r = oracle.plan_informed(seed=s, **kw)
        util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "seed %d" % s)
        assert (out["paths"][i] is None) == (r["path"] is None)
        if r["path"] is not None:
            assert np.array_equal(out["paths"][i], r["path"]) and out["results"][0][i] == r["c_best"]


def test_gpu_informed_candidate_order_does_not_change_the_tree(gpu, monkeypatch):
    """rrt_07 kernel: the default candidate order (choose_parent cheapest first in growing batches, rewire tests only
    candidates whose cost improves) against RRTX_INFORMED_EAGER=1 (every near candidate tested, as the reference does) and
    against the oracle -- C3-style scene, and a cluttered scene where most cheapest candidates are blocked (the batches grow)."""
    import oracle
    g = util.load_golden(util.GOLDEN + "/rrt07_c3_sobol_s1_it3000.npz")
    kw = util.informed_kwargs_from_golden(g)
    kw["max_iter"] = 2500
    import random as _r
    rr = _r.Random(5)
    dense = dict(kw)
    dense["obstacles"] = []
    while len(dense["obstacles"]) < 200:     # 200 circles on a quarter of the C3 map: ~20 % of the area is blocked
        ox, oy, orad = rr.uniform(0, 50), rr.uniform(0, 50), rr.uniform(0.3, 1.5)
        if min((ox - 2) ** 2 + (oy - 2) ** 2, (ox - 48) ** 2 + (oy - 48) ** 2) > (orad + 3) ** 2:
            dense["obstacles"].append((ox, oy, orad))
    dense.update(start=[2.0, 2.0], goal=[48.0, 48.0], rand_area=[0.0, 50.0], expand_dis=1.5, max_iter=2500)
    for name, k in (("c3", kw), ("dense", dense)):
        seeds = list(range(1, 9))
        lazy = util.run_gpu_informed(k, seeds)
        monkeypatch.setenv("RRTX_INFORMED_EAGER", "1")
        eager = util.run_gpu_informed(k, seeds)
        monkeypatch.delenv("RRTX_INFORMED_EAGER")
        for i, s in enumerate(seeds):
            r = oracle.plan_informed(seed=s, **k)
            ref = (r["x"], r["y"], r["cost"], r["parent"])
            util.assert_tree_equal(lazy["trees"][i], ref, "%s default order, seed %d" % (name, s))
            util.assert_tree_equal(eager["trees"][i], ref, "%s eager, seed %d" % (name, s))
            assert (lazy["paths"][i] is None) == (r["path"] is None)
            if r["path"] is not None:
                assert np.array_equal(lazy["paths"][i], r["path"]) and lazy["results"][0][i] == r["c_best"]
        # collision verdicts from the exact (atan2 / cos / sin end point) form for every near obstacle instead of the
        # tolerance band around the new node: same trees (the band only decides which form is evaluated)
        monkeypatch.setenv("RRTX_INFORMED_EXACT_SEG", "1")
        exact = util.run_gpu_informed(k, seeds)
        monkeypatch.delenv("RRTX_INFORMED_EXACT_SEG")
        for i, s in enumerate(seeds):
            util.assert_tree_equal(exact["trees"][i], lazy["trees"][i], "%s exact segment form, seed %d" % (name, s))
        # the reference-equivalent count is the same, the device tests several times fewer segments
        assert lazy["stats"]["edges_ref"] == eager["stats"]["edges_ref"]
        assert lazy["stats"]["edges_unique"] * 2 < eager["stats"]["edges_unique"]
        assert lazy["stats"]["rewires"] == eager["stats"]["rewires"]


def test_gpu_plans_resume_across_kernel_launches(gpu, monkeypatch):
    """A plan is one kernel launch by default (131 072 / 32 768 iterations per launch); longer plans, and RRTX_CHUNK_ITERS, split
    it, and every launch resumes from the state the previous one stored (tree, RNG / Sobol state, prefetched nearest query,
    c_best, polyline pool).  300-iteration launches against the reference goldens, one planner per kernel."""
    monkeypatch.setenv("RRTX_CHUNK_ITERS", "300")
    g = util.load_golden(util.GOLDEN + "/rrt04_c2_s1_it4000.npz")                     # rrt_04 iteration kernel
    out = util.run_gpu_batch(util.kwargs_from_golden(g), [int(g["seed"])])
    _check_against_golden(g, out)
    g = util.load_golden(util.GOLDEN + "/rrt04_c2_sobol_s4_it1500.npz")
    out = util.run_gpu_batch(util.kwargs_from_golden(g), [int(g["seed"])])
    _check_against_golden(g, out)
    assert out["sobol_index"][0] == int(g["sobol_index_after"])
    g = util.load_golden(util.GOLDEN + "/rrt01_drv_s42.npz")                          # general kernel
    _check_against_golden(g, util.run_gpu_batch(util.kwargs_from_golden(g), [int(g["seed"])]))
    g = util.load_golden(util.GOLDEN + "/rrt07_c3_sobol_s1_it3000.npz")               # rrt_07
    out = util.run_gpu_informed(util.informed_kwargs_from_golden(g), [int(g["seed"])])
    util.assert_tree_equal(out["trees"][0], (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    assert (out["paths"][0] is None) == (len(g["path"]) == 0)
    if len(g["path"]):
        assert np.array_equal(out["paths"][0], g["path"]) and out["results"][0][0] == float(g["path_len"])
    g = util.load_golden(util.GOLDEN + "/rrt05_drv_s3_it1500.npz")                    # rrt_05
    out = util.run_gpu_dubins(g, [int(g["seed"])])
    util.assert_tree_equal(out["trees"][0], (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    assert np.array_equal(out["yaws"][0], g["yaw"]) and np.array_equal(out["polys"][0][1], g["poly_x"])
    g = util.load_golden(util.GOLDEN + "/rrt03_drv_s9_it1500_mt.npz")                 # rrt_03 (same kernel, plain mode)
    out = util.run_gpu_rrt_dubins(g, [int(g["seed"])])
    util.assert_tree_equal(out["trees"][0], (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    g = util.load_golden(util.GOLDEN + "/rrt06_drv_s7_it750.npz")                     # rrt_06
    out = util.run_gpu_rrt_rs(g, [int(g["seed"])])
    util.assert_tree_equal(out["trees"][0], (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    if len(g["path"]):
        assert np.array_equal(out["paths"][0], g["path"][:, :2]) and np.array_equal(out["path_yaws"][0], g["path"][:, 2])


@pytest.mark.parametrize("chunk", [None, "300", "97"])
def test_gpu_one_wave_shape_with_shared_passes_matches_the_goldens(gpu, monkeypatch, chunk):
    """The kernel shape the bench times (64 threads per instance, 16-bit stage) answers the near queries of up to three
    iterations from the streaming pass of an earlier one: samples are drawn up to four ahead, so the RNG /
    Sobol state handed back, the per-iteration trace and the counters are the sensitive outputs.  Every rrt_04 golden that
    keeps planning to max_iter, as one launch and as launches of 300 / 97 iterations (the look-ahead must stop at a launch's
    end); with RRTX_SPEC2=0 (one pass per iteration) the same trees."""
    monkeypatch.setenv("RRTX_TPB", "64")
    if chunk:
        monkeypatch.setenv("RRTX_CHUNK_ITERS", chunk)
    rode = 0
    for path in util.golden_files("rrt04_c2") + util.golden_files("rrt04_drv"):
        g = util.load_golden(path)
        kw = util.kwargs_from_golden(g)
        if not kw.get("search_until_max_iter", True):
            continue
        out = util.run_gpu_batch(kw, [int(g["seed"])], trace_instance=0)
        d = util.first_trace_divergence(out["trace"], g["tr_rnd_x"], g["tr_rnd_y"], g["tr_nearest"])
        assert d is None, "%s: first divergent iteration %d" % (g["name"], d)
        _check_against_golden(g, out)
        assert out["stats"]["edges_ref"] == int(g["ref_edges"]), g["name"]
        if kw["sobol"]:
            assert out["sobol_index"][0] == int(g["sobol_index_after"])
        rode += out["stats"]["passes_shared"]
    assert rode > 0
    if chunk is None:
        monkeypatch.setenv("RRTX_SPEC2", "0")
        g = util.load_golden(util.GOLDEN + "/rrt04_c2_s1_it4000.npz")
        out = util.run_gpu_batch(util.kwargs_from_golden(g), [int(g["seed"])])
        _check_against_golden(g, out)
        assert out["stats"]["passes_shared"] == 0


def test_informed_host_class_drop_in(gpu):
    import random
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt07_drv_mt_s42_it2000.npz")
    kw = util.informed_kwargs_from_golden(g)
    random.seed(42)
    rrt = rrt_amd.InformedRRTStar(start=kw["start"], goal=kw["goal"], obstacle_list=kw["obstacles"],
                                  rand_area=kw["rand_area"], expand_dis=0.5, goal_sample_rate=10, max_iter=2000,
                                  sobol_sampler=False)
    path = rrt.informed_rrt_star_search(animation=False)
    assert len(rrt.node_list) == 1397 and len(path) == 13
    assert rrt.get_path_len(path) == 17.33495114327498
    assert rrt.node_list[5].parent == int(g["parent"][5]) and rrt.node_list[0].parent is None
    assert random.getstate()[1][624] == int(g["rng_pos_after"])


@pytest.mark.parametrize("path", util.golden_files("rrt05"), ids=lambda p: p.split("/")[-1][:-4])
def test_gpu_dubins_matches_reference_golden(gpu, path):
    """rrt_05 RRT*-Dubins on the GPU vs the reference goldens: poses, costs, parents, every stored edge polyline,
    the final course and the RNG state (ints exact; doubles bit-exact on this image, contract 1e-6)."""
    g = util.load_golden(path)
    out = util.run_gpu_dubins(g, [int(g["seed"])], trace_instance=0)
    d = util.first_trace_divergence(out["trace"], g["tr_rx"], g["tr_ry"], g["tr_nearest"])
    assert d is None, "first divergent iteration %d" % d
    x, y, cost, parent = out["trees"][0]
    assert len(x) == len(g["x"]) and np.array_equal(parent, g["parent"])
    assert np.allclose(x, g["x"], rtol=0, atol=1e-6) and np.allclose(out["yaws"][0], g["yaw"], rtol=0, atol=1e-6)
    util.assert_tree_equal(out["trees"][0], (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    assert np.array_equal(out["yaws"][0], g["yaw"])
    plen, px, py = out["polys"][0]
    assert np.array_equal(plen, g["poly_len"]) and np.array_equal(px, g["poly_x"]) and np.array_equal(py, g["poly_y"])
    p = out["paths"][0]
    if len(g["path"]) == 0:
        assert p is None
    else:
        assert p is not None and np.array_equal(p, g["path"])
    st = out["rng"][0]
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])


@pytest.mark.parametrize("path", util.golden_files("rrt06"), ids=lambda p: p.split("/")[-1][:-4])
def test_gpu_reeds_shepp_matches_reference_golden(gpu, path):
    """rrt_06 RRT*-Reeds-Shepp on the GPU vs the reference goldens: poses, costs, parents (incl. the try_goal_path
    nodes), every stored edge polyline, the three-column final course and the RNG state (ints exact; doubles
    bit-exact on this image, contract 1e-6)."""
    g = util.load_golden(path)
    out = util.run_gpu_rrt_rs(g, [int(g["seed"])], trace_instance=0)
    d = util.first_trace_divergence(out["trace"], g["tr_rx"], g["tr_ry"], g["tr_nearest"])
    assert d is None, "first divergent iteration %d" % d
    x, y, cost, parent = out["trees"][0]
    assert len(x) == len(g["x"]) and np.array_equal(parent, g["parent"])
    assert np.allclose(x, g["x"], rtol=0, atol=1e-6) and np.allclose(out["yaws"][0], g["yaw"], rtol=0, atol=1e-6)
    util.assert_tree_equal(out["trees"][0], (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    assert np.array_equal(out["yaws"][0], g["yaw"])
    plen, px, py = out["polys"][0]
    assert np.array_equal(plen, g["poly_len"]) and np.array_equal(px, g["poly_x"]) and np.array_equal(py, g["poly_y"])
    p = out["paths"][0]
    if len(g["path"]) == 0:
        assert p is None
    else:
        assert p is not None and np.array_equal(p, g["path"][:, :2])
        assert np.array_equal(out["path_yaws"][0], g["path"][:, 2])
    st = out["rng"][0]
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])


def _orc06(a):
    import oracle
    g, sd, it = a
    r = oracle.plan_rrt_rs(g["start"], g["goal"], g["obstacles"], g["rand_area"], it, seed=sd,
                           curvature=float(g["curvature"]), robot_radius=float(g["robot_radius"]),
                           expand_dis=float(g["expand_dis"]), connect_circle_dist=float(g["connect_circle_dist"]),
                           step_size=float(g["step_size"]))
    return r["x"], r["y"], r["cost"], r["parent"], r["yaw"], r["poly_x"], r["path"], r["path_yaw"], r["stats"]


def test_gpu_reeds_shepp_many_seeds_equal_oracle(gpu):
    """96 seeds x the driver's 750 iterations (rrt_06:2012-2083) in one launch vs the oracle (itself pinned by the
    goldens): trees, yaws, polylines and final courses."""
    from concurrent.futures import ProcessPoolExecutor
    g = {k: v for k, v in util.load_golden(util.GOLDEN + "/rrt06_drv_s7_it750.npz").items()}
    seeds = list(range(100, 196))
    out = util.run_gpu_rrt_rs(g, seeds)
    with ProcessPoolExecutor(max_workers=8) as ex:
        refs = list(ex.map(_orc06, [(g, s, 750) for s in seeds]))
    found = 0
    for i, s in enumerate(seeds):
        r = refs[i]
        util.assert_tree_equal(out["trees"][i], r[:4], "seed %d" % s)
        assert np.array_equal(out["yaws"][i], r[4]) and np.array_equal(out["polys"][i][1], r[5])
        assert (out["paths"][i] is None) == (r[6] is None)
        if r[6] is not None:
            found += 1
            assert np.array_equal(out["paths"][i], r[6]) and np.array_equal(out["path_yaws"][i], r[7])
    assert found > 48
    assert out["stats"]["rewires"] == sum(r[8]["rewires"] for r in refs)
    # default = lazy candidate order (only edges that can change the result are steered); RRTX_RS_EAGER=1 steers every
    # choose_parent / rewire candidate like the reference: same trees, and then the unit of work (collision-checked
    # edges) is the oracle's count exactly
    assert out["stats"]["edges_unique"] < sum(r[8]["edges_unique"] for r in refs)
    import os
    os.environ["RRTX_RS_EAGER"] = "1"
    try:
        eag = util.run_gpu_rrt_rs(g, seeds[:24])
    finally:
        del os.environ["RRTX_RS_EAGER"]
    for i in range(24):
        util.assert_tree_equal(eag["trees"][i], out["trees"][i], "eager vs lazy, seed %d" % seeds[i])
        assert np.array_equal(eag["polys"][i][1], out["polys"][i][1])
    assert eag["stats"]["edges_unique"] == sum(r[8]["edges_unique"] for r in refs[:24])


def test_reeds_shepp_host_class_drop_in(gpu):
    """The rrt_06 driver (:2012-2087) through the drop-in class: same constructor arguments, same `random` stream."""
    import random
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt06_drv_s42_it200.npz")
    random.seed(42)
    rrt = rrt_amd.RRTStarReedsShepp(start=list(g["start"]), goal=list(g["goal"]),
                                    obstacle_list=[tuple(o) for o in g["obstacles"]], rand_area=list(g["rand_area"]),
                                    expand_dis=3.0, path_resolution=0.5, goal_sample_rate=10, max_iter=200,
                                    play_area=None, robot_radius=0.6, sobol_sampler=True, connect_circle_dist=50.0,
                                    search_until_max_iter=False, curvature=2.0, goal_yaw_th=float(np.deg2rad(1.0)),
                                    goal_xy_th=0.5, step_size=0.1)
    path = rrt.planning(animation=False)
    assert path is not None and np.array_equal(np.array(path), g["path"])
    assert len(rrt.node_list) == len(g["x"]) and rrt.node_list[3].cost == float(g["cost"][3])
    assert random.getstate()[1][624] == int(g["rng_pos_after"])
    smoothed = rrt_amd.path_smoothing([[a, b] for a, b, _ in path], 100, [tuple(o) for o in g["obstacles"]])
    assert len(smoothed) >= 2 and smoothed[0] == [float(g["goal"][0]), float(g["goal"][1])]


def test_batch_planner_pose_planners(gpu):
    """BatchPlanner over the pose planners: instance i = the single-instance class seeded with seeds[i]."""
    import random
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt06_drv_s42_it200.npz")
    obst = [tuple(o) for o in g["obstacles"]]
    bp = rrt_amd.BatchPlanner("rrt_star_reeds_shepp", [42, 43, 44], list(g["start"]), list(g["goal"]), obst,
                              list(g["rand_area"]), expand_dis=3.0, goal_sample_rate=10, max_iter=200, robot_radius=0.6,
                              search_until_max_iter=True, curvature=2.0, step_size=0.1)
    try:
        bp.plan()
        assert np.array_equal(bp.path(0), g["path"]) and np.array_equal(bp.yaw(0), g["yaw"])
        random.seed(44)
        one = rrt_amd.RRTStarReedsShepp(list(g["start"]), list(g["goal"]), obst, list(g["rand_area"]), goal_sample_rate=10,
                                        max_iter=200, robot_radius=0.6, curvature=2.0, step_size=0.1)
        p = one.planning(animation=False)
        assert (p is None) == (bp.path(2) is None) and (p is None or np.array_equal(np.array(p), bp.path(2)))
        assert np.array_equal(bp.tree(2)[2], one.tree[2])
    finally:
        bp.close()
    g5 = util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it150.npz")
    bp = rrt_amd.BatchPlanner("rrt_star_dubins", [42, 7], list(g5["start"]), list(g5["goal"]),
                              [tuple(o) for o in g5["obstacles"]], list(g5["rand_area"]), goal_sample_rate=10, max_iter=150,
                              search_until_max_iter=True, curvature=1.0)
    try:
        bp.plan()
        util.assert_tree_equal(bp.tree(0), (g5["x"], g5["y"], g5["cost"], g5["parent"]), "rrt05 via BatchPlanner")
        assert np.array_equal(bp.polylines(0)[1], g5["poly_x"])
    finally:
        bp.close()


def test_gpu_dubins_batch_equals_oracle(gpu):
    import oracle
    g = util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it500.npz")
    seeds = list(range(1, 9))
    out = util.run_gpu_dubins(g, seeds, max_iter=400)
    for i, s in enumerate(seeds):
        r = oracle.plan_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], 400, seed=s)
        util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "seed %d" % s)
        assert np.array_equal(out["yaws"][i], r["yaw"])
        assert np.array_equal(out["polys"][i][1], r["poly_x"])
        assert (out["paths"][i] is None) == (r["path"] is None)
        if r["path"] is not None:
            assert np.array_equal(out["paths"][i], r["path"])


def _orc05(a):
    import oracle
    g, sd, it = a
    r = oracle.plan_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], it, seed=sd)
    return r["x"], r["y"], r["cost"], r["parent"]


def test_gpu_dubins_many_seeds_equal_oracle(gpu):
    """64 seeds x 3 000 iterations: long enough for the rare paths (e.g. a node rewired twice in one iteration because
    equal-distance nodes collapse onto it in near_inds, rrt_05:1737-1738 -- 15 % of seeds hit that by 3 000 iterations)."""
    from concurrent.futures import ProcessPoolExecutor
    g = {k: v for k, v in util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it500.npz").items()}
    seeds = list(range(330, 394))
    out = util.run_gpu_dubins(g, seeds, max_iter=3000)
    with ProcessPoolExecutor(max_workers=8) as ex:
        refs = list(ex.map(_orc05, [(g, s, 3000) for s in seeds]))
    for i, s in enumerate(seeds):
        util.assert_tree_equal(out["trees"][i], refs[i], "seed %d" % s)


def test_gpu_dubins_lazy_candidate_order(gpu):
    """RRTX_DUBINS_LAZY=1 (opt-in): the rrt_05 kernel with the lazy candidate order of the rrt_06 kernel builds the
    goldens' trees (incl. the run where a node is rewired twice in one iteration) and the oracle's for 48 long runs."""
    import os
    from concurrent.futures import ProcessPoolExecutor
    os.environ["RRTX_DUBINS_LAZY"] = "1"
    try:
        for path in util.golden_files("rrt05_drv"):
            g = util.load_golden(path)
            out = util.run_gpu_dubins(g, [int(g["seed"])])
            util.assert_tree_equal(out["trees"][0], (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
            assert np.array_equal(out["yaws"][0], g["yaw"]) and np.array_equal(out["polys"][0][1], g["poly_x"])
            if len(g["path"]):
                assert np.array_equal(out["paths"][0], g["path"])
        g = {k: v for k, v in util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it500.npz").items()}
        seeds = list(range(360, 408))
        out = util.run_gpu_dubins(g, seeds, max_iter=3000)
        lazy_edges = out["stats"]["edges_unique"]
    finally:
        del os.environ["RRTX_DUBINS_LAZY"]
    with ProcessPoolExecutor(max_workers=8) as ex:
        refs = list(ex.map(_orc05, [(g, s, 3000) for s in seeds]))
    for i, s in enumerate(seeds):
        util.assert_tree_equal(out["trees"][i], refs[i], "seed %d" % s)
    # default = filtered candidate stages; RRTX_DUBINS_FILTER=0 = every candidate steered like the reference
    flt = util.run_gpu_dubins(g, seeds[:8], max_iter=3000)
    os.environ["RRTX_DUBINS_FILTER"] = "0"
    try:
        eager = util.run_gpu_dubins(g, seeds[:8], max_iter=3000)
    finally:
        del os.environ["RRTX_DUBINS_FILTER"]
    for i in range(8):
        util.assert_tree_equal(flt["trees"][i], refs[i], "filtered, seed %d" % seeds[i])
        util.assert_tree_equal(eager["trees"][i], refs[i], "unfiltered, seed %d" % seeds[i])
        assert np.array_equal(flt["polys"][i][1], eager["polys"][i][1])
    assert flt["stats"]["edges_unique"] * 2 < eager["stats"]["edges_unique"]
    assert lazy_edges / 48 < eager["stats"]["edges_unique"] / 8 / 4


def test_dubins_host_class_drop_in(gpu):
    import random
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it150.npz")
    random.seed(42)
    rrt = rrt_amd.RRTStarDubins(start=list(g["start"]), goal=list(g["goal"]),
                                obstacle_list=[tuple(o) for o in g["obstacles"]], rand_area=list(g["rand_area"]),
                                expand_dis=3.0, path_resolution=0.5, goal_sample_rate=10, max_iter=150,
                                robot_radius=0.0, sobol_sampler=True, connect_circle_dist=50.0,
                                search_until_max_iter=True, curvature=1.0)
    path = rrt.planning(animation=False)
    assert path is None and len(rrt.node_list) == 27
    assert max(nd.cost for nd in rrt.node_list) == 8.634402421885998
    assert random.getstate()[1][624] == int(g["rng_pos_after"])


def _bit_coords(ids, rand_min=-2.0):
    c1 = np.floor(ids / 1700.0)
    c0 = np.floor((ids - c1 * 1700.0) / 1)
    return rand_min + 0.01 * c0, rand_min + 0.01 * c1


@pytest.mark.parametrize("path", util.golden_files("rrt03"), ids=lambda p: p.split("/")[-1][:-4])
def test_gpu_rrt_dubins_matches_reference_golden(gpu, path):
    """rrt_03 (RRT with Dubins steer, MT and 3-D Sobol samplers): poses, Dubins-length costs, parents, stored edge
    polylines, final course, RNG state and Sobol index equal the reference's, bit for bit."""
    g = util.load_golden(path)
    out = util.run_gpu_rrt_dubins(g, [int(g["seed"])], trace_instance=0)
    x, y, cost, parent = out["trees"][0]
    tr = out["trace"]
    n = len(g["tr_nearest"])
    assert np.array_equal(tr[0][:n], g["tr_rx"]) and np.array_equal(tr[2][:n], g["tr_nearest"])
    util.assert_tree_equal((x, y, cost, parent), (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    assert np.array_equal(out["yaws"][0], g["yaw"])
    plen, px, py = out["polys"][0]
    assert np.array_equal(plen, g["poly_len"]) and np.array_equal(px, g["poly_x"]) and np.array_equal(py, g["poly_y"])
    p = out["paths"][0]
    if len(g["path"]) == 0:
        assert p is None
    else:
        assert p is not None and np.array_equal(p, g["path"])
    st = out["rng"][0]
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])
    if int(g["sobol"]):
        assert out["sobol"][0] == int(g["sobol_index_after"])


def test_gpu_rrt_dubins_batch_equals_oracle(gpu):
    """rrt_03 batch: instance i == oracle(seed i) at 2 000 iterations (beyond the goldens), both samplers."""
    import oracle
    g = util.load_golden(util.GOLDEN + "/rrt03_drv_s42_it200_sobol.npz")
    for sob in (1, 0):
        g2 = dict(g)
        g2["max_iter"] = 2000
        g2["sobol"] = sob
        seeds = list(range(1, 17))
        out = util.run_gpu_rrt_dubins(g2, seeds)
        for i, sd in enumerate(seeds):
            r = oracle.plan_rrt_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], 2000, seed=sd,
                                       robot_radius=float(g["robot_radius"]), goal_sample_rate=int(g["goal_sample_rate"]),
                                       sobol=bool(sob))
            util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "seed %d" % sd)
            assert np.array_equal(out["yaws"][i], r["yaw"])
            p = out["paths"][i]
            assert (p is None and r["path"] is None) or np.array_equal(p, r["path"])


def test_dubins_host_classes_early_return(gpu):
    """planning(animation=False, search_until_max_iter=False) of rrt_05 and rrt_03 (:1443-1446): returns at the first
    iteration whose tree holds a goal node; tree, path and RNG state as the reference leaves them."""
    import random
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt05_early_s3_it1500.npz")
    obst = [tuple(float(v) for v in o) for o in g["obstacles"]]
    random.seed(int(g["seed"]))
    rrt = rrt_amd.RRTStarDubins(start=[float(v) for v in g["start"]], goal=[float(v) for v in g["goal"]],
                                obstacle_list=obst, rand_area=[float(v) for v in g["rand_area"]], expand_dis=3.0,
                                path_resolution=0.5, goal_sample_rate=10, max_iter=int(g["max_iter"]), robot_radius=0.0,
                                connect_circle_dist=50.0, curvature=1.0)
    path = rrt.planning(animation=False, search_until_max_iter=False)
    assert np.array_equal(np.array(path), g["path"]) and len(rrt.node_list) == len(g["x"])
    st = random.getstate()
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])
    g = util.load_golden(util.GOLDEN + "/rrt03_early_s1_it1500_mt.npz")
    random.seed(int(g["seed"]))
    rrt = rrt_amd.RRTDubins(start=[float(v) for v in g["start"]], goal=[float(v) for v in g["goal"]], obstacle_list=obst,
                            rand_area=[float(v) for v in g["rand_area"]], goal_sample_rate=10,
                            max_iter=int(g["max_iter"]), robot_radius=0.6, sobol_sampler=False, curvature=1.0)
    path = rrt.planning(animation=False, search_until_max_iter=False)
    assert np.array_equal(np.array(path), g["path"]) and len(rrt.node_list) == len(g["x"])
    st = random.getstate()
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])


def test_rrt_dubins_host_class_drop_in(gpu):
    import random
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt03_drv_s42_it200_sobol.npz")
    random.seed(int(g["seed"]))
    rrt = rrt_amd.RRTDubins(start=[float(v) for v in g["start"]], goal=[float(v) for v in g["goal"]],
                            obstacle_list=[tuple(float(v) for v in o) for o in g["obstacles"]],
                            rand_area=[float(v) for v in g["rand_area"]], goal_sample_rate=10, max_iter=200,
                            play_area=None, robot_radius=0.6, sobol_sampler=True, curvature=1.0,
                            goal_yaw_th=float(np.deg2rad(1.0)), goal_xy_th=0.5)
    path = rrt.planning(animation=False)
    assert path is not None and np.array_equal(np.array(path), g["path"])
    assert len(rrt.node_list) == len(g["x"]) and rrt.node_list[-1].cost == float(g["cost"][-1])
    assert rrt.node_list[5].parent is rrt.node_list[int(g["parent"][5])]
    assert rrt.sobol_inter_ == int(g["sobol_index_after"])
    st = random.getstate()
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])


@pytest.mark.parametrize("path", util.golden_files("rrt08"), ids=lambda p: p.split("/")[-1][:-4])
def test_gpu_bitstar_matches_reference_golden(gpu, path):
    """rrt_08 BIT* on the GPU vs the reference goldens: popped-edge sequence, tree vertices (grid coordinates),
    g-scores, parents, returned path (start -> goal) and RNG state."""
    g = util.load_golden(path)
    out = util.run_gpu_bitstar([tuple(float(v) for v in o) for o in g["obstacles"]], [float(v) for v in g["rand_area"]],
                               int(g["max_iter"]), [int(g["seed"])], [[float(v) for v in g["start"]]],
                               [[float(v) for v in g["goal"]]], trace_instance=0)
    tr = out["trace"]
    assert np.array_equal(tr[0], g["tr_e0"]) and np.array_equal(tr[1], g["tr_e1"])
    x, y, cost, parent = out["trees"][0]
    gx, gy = _bit_coords(g["vertex_ids"])
    assert len(x) == len(gx) and np.array_equal(x, gx) and np.array_equal(y, gy) and np.array_equal(cost, g["g_scores"])
    want_parent = np.array([-1 if p < 0 else int(np.nonzero(g["vertex_ids"] == p)[0][0]) for p in g["parent_ids"]])
    assert np.array_equal(parent, want_parent)
    p = out["paths"][0]
    if len(g["path"]) == 0:
        assert p is None
    else:
        assert np.array_equal(p, g["path"])
    st = out["rng"][0]
    assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])


@pytest.mark.parametrize("kernel,n", [("wave", 160), ("lane", 12)])
def test_gpu_bitstar_batch_c4_style_equals_oracle(gpu, monkeypatch, kernel, n):
    """C4-style batch: per-instance start/goal from random.Random(2000+i) in [-1,14]^2 outside the
    obstacles, planner seed 1000+i; every instance equals the oracle's run.  Both device kernels: one wave per
    instance (default) and one lane per instance (the fallback for problems whose vertex state exceeds LDS)."""
    import random
    import oracle
    if kernel == "lane":
        monkeypatch.setenv("RRTX_BITSTAR", "lane")
    obst = [(5, 5, 0.5), (9, 6, 1), (7, 5, 1), (1, 5, 1), (3, 6, 1), (7, 9, 1)]

    def free_point(rng):
        while True:
            x, y = rng.uniform(-1, 14), rng.uniform(-1, 14)
            if all((x - ox) ** 2 + (y - oy) ** 2 > r ** 2 for ox, oy, r in obst):
                return [x, y]
    starts, goals, seeds = [], [], []
    for i in range(n):
        rng = random.Random(2000 + i)
        starts.append(free_point(rng))
        goals.append(free_point(rng))
        seeds.append(1000 + i)
    out = util.run_gpu_bitstar(obst, [-2.0, 15.0], 80, seeds, starts, goals)
    for i in range(n):
        r = oracle.plan_bitstar(starts[i], goals[i], obst, [-2, 15], 80, seed=seeds[i])
        x, y, cost, parent = out["trees"][i]
        gx, gy = _bit_coords(r["vertex_ids"])
        assert np.array_equal(x, gx) and np.array_equal(y, gy) and np.array_equal(cost, r["g_scores"]), i
        p = out["paths"][i]
        assert (p is None and len(r["path"]) == 0) or np.array_equal(p, r["path"]), i


def test_gpu_input_limits(gpu):
    """Maximum sizes and refused inputs: 256 circles (the device tile's capacity) plan as the oracle does, in the
    iteration kernel's widest shape and in the general kernel; 257 are refused by rrtx_set_obstacles, as are a handle of
    0 instances, a negative max_iter and an instance number outside the batch -- errors, never a clipped input."""
    import rrt_amd
    A = rrt_amd._abi
    kw = util.c2_kwargs(1500, m=256, map_seed=13)
    r = util.run_oracle(kw, 5, exact_pow=True)
    for until in (1, 0):
        kw2 = dict(kw, search_until_max_iter=until)
        ro = r if until else util.run_oracle(kw2, 5, exact_pow=True)
        out = util.run_gpu_batch(kw2, [5, 6])
        util.assert_tree_equal(out["trees"][0], (ro["x"], ro["y"], ro["cost"], ro["parent"]), "256 obstacles, until_max %d" % until)
    kw3 = util.c2_kwargs(10, m=257, map_seed=13)
    with pytest.raises(A.RrtxError):
        util.run_gpu_batch(kw3, [5])
    common = ([2.0, 2.0], [98.0, 98.0], [0, 100], 2.0, 0.25, 5)
    with pytest.raises(A.RrtxError):
        A.Handle(A.ALGO_RRT_STAR, *common, 100, n_instances=0)
    with pytest.raises(A.RrtxError):
        A.Handle(A.ALGO_RRT_STAR, *common, -1, n_instances=1)
    h = A.Handle(A.ALGO_RRT_STAR, *common, 100, n_instances=2)
    try:
        with pytest.raises(A.RrtxError):
            h.set_instance(2, [2.0, 2.0], [98.0, 98.0])
        with pytest.raises(A.RrtxError):
            h.get_tree(0)   # nothing planned yet
    finally:
        h.close()


@pytest.mark.timeout(180)
def test_gpu_bitstar_start_inside_an_obstacle_ends_as_overflow(gpu):
    """A start inside a circle: every connect() of rrt_08's plan() fails, its `continue` (:283) skips the iteration
    counter, and because samples are only added `if iterations != 0` (:215) the reference repeats the same round for ever
    (measured: no return within 20 minutes).  The device proves that at the second time both queues run dry with nothing
    changed and stops the instance at once with RRTX_ST_OVERFLOW | RRTX_ST_REF_HANGS (round 2: after 4 million trips, 18 s
    on one wave; the oracle gives up at its own trip guard), the call returns RRTX_PARTIAL, and the other instance of the
    batch is complete and equal to the oracle."""
    import oracle
    import rrt_amd
    A = rrt_amd._abi
    obst = [(5, 5, 0.5), (9, 6, 1), (7, 5, 1), (1, 5, 1), (3, 6, 1), (7, 9, 1)]
    starts, goals, seeds = [[7.0, 5.0], [-1.0, 0.0]], [[3.0, 8.0], [3.0, 8.0]], [3, 42]
    with pytest.raises(RuntimeError):
        oracle.plan_bitstar(starts[0], goals[0], obst, [-2, 15], 40, seed=3)
    c_min, c = rrt_amd.bitstar_rotation(starts[0], goals[0])
    h = A.Handle(A.ALGO_BITSTAR, starts[0], goals[0], [-2.0, 15.0], 2.0, 1.0, 0, 40, n_instances=2,
                 informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
    try:
        h.set_obstacles(obst)
        h.seed_instances(seeds)
        for i in range(2):
            cm, ci = rrt_amd.bitstar_rotation(starts[i], goals[i])
            h.set_instance(i, starts[i], goals[i])
            h.set_instance_rotation(i, [ci[0, 0], ci[0, 1], ci[1, 0], ci[1, 1]], cm)
        import time
        t0 = time.perf_counter()
        h.plan(strict=False)
        assert time.perf_counter() - t0 < 2.0
        st = h.get_results()[2]
        assert st[0] & A.ST_OVERFLOW and st[0] & A.ST_REF_HANGS and not (st[1] & A.ST_FAILED)
        r = oracle.plan_bitstar(starts[1], goals[1], obst, [-2, 15], 40, seed=42)
        x, y, cost, parent = h.get_tree(1)
        gx, gy = _bit_coords(r["vertex_ids"])
        assert np.array_equal(x, gx) and np.array_equal(y, gy) and np.array_equal(cost, r["g_scores"])
    finally:
        h.close()


def test_bitstar_host_class_drop_in(gpu):
    import random
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt08_s42_it80.npz")
    random.seed(42)
    b = rrt_amd.BITStar(start=[-1.0, 0.0], goal=[3.0, 8.0], obstacleList=[tuple(o) for o in g["obstacles"]],
                        randArea=[-2, 15], maxIter=80, lowerLimit=[0.0, 0.0], upperLimit=[0.0, 10.0], resolution=1.0,
                        eta=2.0)
    path = b.plan(animation=False)
    assert len(path) == 9 and np.array_equal(np.array(path), g["path"])
    assert random.getstate()[1][624] == int(g["rng_pos_after"])


# ------------------------------------------------------------------------------------------------ path smoothing
def test_gpu_path_smoothing_matches_reference_golden(gpu):
    """path_smoothing (rrt_04:1447-1479) through rrtx_smooth_paths, all goldens as one batch per max_iter: smoothed
    polyline and MT19937 state after the call, bit for bit."""
    import rrt_amd
    gs = [util.load_golden(p) for p in util.golden_files("smooth")]
    assert len(gs) >= 6
    for it in sorted({int(g["max_iter"]) for g in gs}):
        for obst_key in sorted({g["obstacles"].tobytes() for g in gs if int(g["max_iter"]) == it}):
            grp = [g for g in gs if int(g["max_iter"]) == it and g["obstacles"].tobytes() == obst_key]
            out, states, st = rrt_amd._abi.smooth_paths([g["path_in"] for g in grp], it, grp[0]["obstacles"],
                                                        [(g["rng_mt_before"], int(g["rng_pos_before"])) for g in grp])
            for g, o, (w, pos) in zip(grp, out, states):
                assert np.array_equal(o, g["smoothed"]), g["name"]
                assert pos == int(g["rng_pos_after"]) and int(w[0]) == int(g["rng_word0_after"]), g["name"]


def test_driver_sequence_planning_then_smoothing(gpu):
    """The reference driver's sequence (rrt_04:1548-1559): random.seed -> RRT(...).planning() -> path_smoothing(path, 1000,
    obstacleList), all on the global random stream; and the same as a device-resident batch (rrtx_smooth_planned)."""
    import random
    import rrt_amd
    obst = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)]
    seeds = [1234, 5, 7]
    gold = {sd: util.load_golden(util.GOLDEN + "/smooth_drv_s%d.npz" % sd) for sd in seeds}
    for sd in seeds:
        g = gold[sd]
        random.seed(sd)
        rrt = rrt_amd.RRTStar(start=[0, 0], goal=[6.0, 10.0], obstacle_list=obst, rand_area=[-2, 15], expand_dis=1.0,
                              path_resolution=0.1, goal_sample_rate=5, max_iter=500, play_area=[0, 10, 0, 14],
                              robot_radius=0.6, sobol_sampler=False, connect_circle_dist=50.0, search_until_max_iter=True)
        path = rrt.planning(animation=False)
        assert np.array_equal(np.array(path), g["path_in"])
        sm = rrt_amd.path_smoothing(path, 1000, obst)
        assert np.array_equal(np.array(sm), g["smoothed"])
        st = random.getstate()
        assert st[1][624] == int(g["rng_pos_after"]) and st[1][0] == int(g["rng_word0_after"])
    A = rrt_amd._abi
    h = A.Handle(A.ALGO_RRT_STAR, [0, 0], [6.0, 10.0], [-2, 15], 1.0, 0.1, 5, 500, play_area=[0, 10, 0, 14],
                 robot_radius=0.6, connect_circle_dist=50.0, search_until_max_iter=True, n_instances=len(seeds))
    try:
        h.set_obstacles(obst)
        h.seed_instances(seeds)
        h.plan()
        h.smooth_planned(1000)
        for i, sd in enumerate(seeds):
            assert np.array_equal(h.get_smoothed_path(i), gold[sd]["smoothed"])
            st = h.get_rng_state(i)
            assert st[1][624] == int(gold[sd]["rng_pos_after"]) and st[1][0] == int(gold[sd]["rng_word0_after"])
    finally:
        h.close()


def test_batch_planner_smooth_and_export(gpu, tmp_path):
    """BatchPlanner: device-resident smoothing of every planned path and the NPZ tree export."""
    import rrt_amd
    obst = [(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)]
    bp = rrt_amd.BatchPlanner("rrt_star", [1234, 5, 7], [0, 0], [6.0, 10.0], obst, [-2, 15], expand_dis=1.0,
                              path_resolution=0.1, goal_sample_rate=5, max_iter=500, play_area=[0, 10, 0, 14],
                              robot_radius=0.6, connect_circle_dist=50.0, search_until_max_iter=True)
    try:
        bp.plan()
        f = bp.export_npz(str(tmp_path / "trees.npz"))
        z = np.load(f)
        g = util.load_golden(util.GOLDEN + "/smooth_drv_s1234.npz")
        assert np.array_equal(z["path_0"], g["path_in"]) and len(z["x_0"]) == int(z["n_nodes"][0])
        sm = bp.smooth(1000)
        assert np.array_equal(sm[0], g["smoothed"])
    finally:
        bp.close()


# ---------------------------------------------------------------- batch surface: per-instance problems (round-2)
@pytest.mark.gpu
def test_batch_planner_informed_per_instance_start_goal(gpu):
    """BatchPlanner("informed"): every instance has its own start / goal, hence its own rotation C and c_min
    (rrt_07:1054-1068, computed on the host with numpy as the reference does); instance i equals the oracle's run of that
    problem, and instance 0 (the rrt_07 driver's problem, seed 42) equals the reference golden."""
    import oracle
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt07_drv_mt_s42_it2000.npz")
    kw = util.informed_kwargs_from_golden(g)
    it = int(g["max_iter"])
    starts = [kw["start"], [0.0, 0.0], [12.0, 0.0], [-1.0, 14.0], [1.0, 9.0]]
    goals = [kw["goal"], [12.0, 12.0], [0.0, 13.0], [12.0, 0.0], [9.0, 1.0]]
    seeds = [int(g["seed"]), 5, 6, 9, 5]
    bp = rrt_amd.BatchPlanner("informed", seeds, kw["start"], kw["goal"], kw["obstacles"], kw["rand_area"], expand_dis=0.5,
                              goal_sample_rate=10, max_iter=it, sobol_sampler=False, starts=starts, goals=goals)
    try:
        pc, nn, st = bp.plan()
        # instance 4 starts in a pocket: its near sets (uncapped radius, rrt_07:1139) outgrow the product shape's 512 LDS
        # candidate slots (~600 of its ~750 nodes); it is planned again on the 2048-slot shape (rrtx_api.hip)
        assert not bp.partial and bp.failed() == []
        util.assert_tree_equal(bp.tree(0), (g["x"], g["y"], g["cost"], g["parent"]), "instance 0 = golden")
        assert (bp.path(0) is None) == (len(g["path"]) == 0)
        if len(g["path"]):
            assert np.array_equal(bp.path(0), g["path"]) and pc[0] == float(g["path_len"])
        for i in range(1, 5):
            k2 = dict(kw)
            k2.update(start=starts[i], goal=goals[i], max_iter=it)
            r = oracle.plan_informed(seed=seeds[i], **k2)
            util.assert_tree_equal(bp.tree(i), (r["x"], r["y"], r["cost"], r["parent"]), "informed instance %d" % i)
            assert (bp.path(i) is None) == (r["path"] is None)
            if r["path"] is not None:
                assert np.array_equal(bp.path(i), r["path"]) and pc[i] == r["c_best"]
    finally:
        bp.close()


@pytest.mark.gpu
def test_batch_planner_bitstar_per_instance_start_goal(gpu):
    """BatchPlanner("bitstar") = BASELINE config C4's shape: per-instance start / goal; instance i equals the single
    class `BITStar` seeded the same way, and the oracle."""
    import random
    import oracle
    import rrt_amd
    obst = [(5, 5, 0.5), (9, 6, 1), (7, 5, 1), (1, 5, 1), (3, 6, 1), (7, 9, 1)]

    def free_point(rng):
        while True:
            x, y = rng.uniform(-1, 14), rng.uniform(-1, 14)
            if all((x - ox) ** 2 + (y - oy) ** 2 > r ** 2 for ox, oy, r in obst):
                return [x, y]
    n = 24
    starts, goals, seeds = [], [], []
    for i in range(n):
        rng = random.Random(2000 + i)
        starts.append(free_point(rng))
        goals.append(free_point(rng))
        seeds.append(1000 + i)
    bp = rrt_amd.BatchPlanner("bitstar", seeds, starts[0], goals[0], obst, [-2.0, 15.0], max_iter=80, starts=starts,
                              goals=goals)
    try:
        pc, nn, st = bp.plan()
        for i in range(n):
            r = oracle.plan_bitstar(starts[i], goals[i], obst, [-2, 15], 80, seed=seeds[i])
            p = bp.path(i)
            assert (p is None and len(r["path"]) == 0) or np.array_equal(p, r["path"]), i
            assert np.array_equal(bp.tree(i)[2], r["g_scores"]), i
        random.seed(seeds[3])
        one = rrt_amd.BITStar(starts[3], goals[3], obst, [-2.0, 15.0], maxIter=80)
        p1 = one.plan(animation=False)
        p3 = bp.path(3)
        assert (p3 is None and p1 == []) or np.array_equal(np.array(p1), p3)
    finally:
        bp.close()


@pytest.mark.gpu
def test_batch_planner_pose_per_instance_yaw(gpu):
    """Per-instance start and goal POSES (x, y, yaw) for the pose planners (round-1 VERDICT missing 3): every instance
    equals the oracle's run of that problem -- rrt_05 (Dubins), rrt_03 (RRT-Dubins) and rrt_06 (Reeds-Shepp)."""
    import oracle
    import rrt_amd
    g5 = util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it150.npz")
    obst = [tuple(o) for o in g5["obstacles"]]
    ra = list(g5["rand_area"])
    starts = [list(g5["start"]), [0.0, 1.0, 0.7], [1.0, -1.0, -2.1], [12.0, 0.0, 3.0]]
    goals = [list(g5["goal"]), [11.0, 9.0, 1.2], [10.0, 12.0, -0.4], [0.0, 12.0, 2.2]]
    seeds = [42, 3, 4, 5]
    bp = rrt_amd.BatchPlanner("rrt_star_dubins", seeds, starts[0], goals[0], obst, ra, goal_sample_rate=10, max_iter=300,
                              search_until_max_iter=True, curvature=1.0, starts=starts, goals=goals)
    try:
        bp.plan()
        for i in range(4):
            r = oracle.plan_dubins(starts[i], goals[i], obst, ra, 300, seed=seeds[i])
            util.assert_tree_equal(bp.tree(i), (r["x"], r["y"], r["cost"], r["parent"]), "rrt05 pose instance %d" % i)
            assert np.array_equal(bp.yaw(i), r["yaw"])
            assert (bp.path(i) is None) == (r["path"] is None)
            if r["path"] is not None:
                assert np.array_equal(bp.path(i), r["path"])
    finally:
        bp.close()
    bp = rrt_amd.BatchPlanner("rrt_dubins", seeds, starts[0], goals[0], obst, ra, goal_sample_rate=10, max_iter=300,
                              search_until_max_iter=True, curvature=1.0, starts=starts, goals=goals)
    try:
        bp.plan()
        for i in range(4):
            r = oracle.plan_rrt_dubins(starts[i], goals[i], obst, ra, 300, seed=seeds[i])
            util.assert_tree_equal(bp.tree(i), (r["x"], r["y"], r["cost"], r["parent"]), "rrt03 pose instance %d" % i)
            assert np.array_equal(bp.yaw(i), r["yaw"])
    finally:
        bp.close()
    g6 = util.load_golden(util.GOLDEN + "/rrt06_drv_s42_it200.npz")
    obst6 = [tuple(o) for o in g6["obstacles"]]
    starts6 = [list(g6["start"]), [0.0, 1.0, 0.7], [1.0, -1.0, -2.1]]
    goals6 = [list(g6["goal"]), [11.0, 12.0, 1.2], [12.0, 2.0, -0.4]]
    bp = rrt_amd.BatchPlanner("rrt_star_reeds_shepp", [42, 8, 9], starts6[0], goals6[0], obst6, list(g6["rand_area"]),
                              expand_dis=3.0, goal_sample_rate=10, max_iter=200, robot_radius=0.6,
                              search_until_max_iter=True, curvature=2.0, step_size=0.1, starts=starts6, goals=goals6)
    try:
        bp.plan()
        assert np.array_equal(bp.path(0), g6["path"])
        for i in range(3):
            r = oracle.plan_rrt_rs(starts6[i], goals6[i], obst6, list(g6["rand_area"]), 200, seed=[42, 8, 9][i],
                                   curvature=2.0, robot_radius=0.6, step_size=0.1)
            util.assert_tree_equal(bp.tree(i), (r["x"], r["y"], r["cost"], r["parent"]), "rrt06 pose instance %d" % i)
            assert np.array_equal(bp.yaw(i), r["yaw"])
            p = bp.path(i)
            assert (p is None) == (r["path"] is None)
            if p is not None:
                assert np.array_equal(p[:, :2], r["path"]) and np.array_equal(p[:, 2], r["path_yaw"])
    finally:
        bp.close()


@pytest.mark.parametrize("res,rate,scene,seed", [(0.05, 60, "diag", 5), (0.05, 95, "diag", 8), (0.1, 20, "drv", 20),
                                                 (0.3, 20, "drv", 30), (0.05, 20, "diag", 19507)])
def test_gpu_rewire_moved_node_equals_oracle(gpu, res, rate, scene, seed):
    """rrt_04 rewire's rare branch: steer(new -> near, inf) stops short of the node (rounding drift of the accumulated
    steps with a path_resolution that is not a binary fraction), so `node_list[i] = edge_node` MOVES the node (:1372).
    tools/find_moved_node.py found these problems with the oracle's counters (moved 17 / 15 / 9 / 3 nodes in 400
    iterations); the device must move the same nodes: trees equal bit for bit, no instance left as UNSUPPORTED.
    When such a rewire falls into an iteration whose near_inds holds repeated indices (goal duplicates, :1337) the
    iteration kernel hands the instance to the general kernel, which walks the raw list visit by visit
    (rppk::rewire_raw_walk): `replanned` counts those instances -- the goal-rate-95 problem is one.  Seed 19507 is the
    one plan in 240 000 (tools/find_moved_node.py, 12 configurations x 20 000 seeds) where the reference really visits a
    moved node a second time -- the first goal node, listed again for every goal duplicate -- and steers to where it lies
    now; the raw-list walk follows it."""
    import ctypes as C
    import oracle
    kw = dict(util.C2)
    if scene == "diag":
        kw.update(start=[0, 0], goal=[6, 8], rand_area=[-2, 12], obstacles=[(3, 3, 1)])
    else:
        kw.update(start=[0, 0], goal=[6, 10], rand_area=[-2, 15],
                  obstacles=[(5, 5, 1), (3, 6, 2), (3, 8, 2), (3, 10, 2), (7, 5, 2), (9, 5, 2), (8, 10, 1)])
    kw.update(expand_dis=3.0, path_resolution=res, goal_sample_rate=rate, connect_circle_dist=50.0, max_iter=400,
              robot_radius=0.0)
    L = oracle.lib()
    m0, r0, m1, r1 = C.c_long(), C.c_long(), C.c_long(), C.c_long()
    L.orc_moved_counters(C.byref(m0), C.byref(r0))
    r = util.run_oracle(kw, seed, exact_pow=True)
    L.orc_moved_counters(C.byref(m1), C.byref(r1))
    assert m1.value - m0.value > 0      # the branch is reached
    out = util.run_gpu_batch(kw, [seed, seed + 1000])
    revisits = r1.value - r0.value
    assert (revisits > 0) == (seed == 19507)   # the one problem where the reference visits a moved node AGAIN (:1337)
    if rate == 95 or revisits:
        assert out["stats"]["replanned"] >= 1
    util.assert_tree_equal(out["trees"][0], (r["x"], r["y"], r["cost"], r["parent"]), "moved-node problem")
    assert (out["paths"][0] is None) == (r["path"] is None)
    if r["path"] is not None:
        assert np.array_equal(out["paths"][0], r["path"])
    assert not (out["results"][2] & 16).any()


@pytest.mark.parametrize("name", ["nodepaths_rrt01_s42", "nodepaths_rrt04_s1234_full", "nodepaths_rrt04_s7_full",
                                  "nodepaths_rrt04_s3_early", "nodepaths_rrt04_res03_s5"])
def test_node_path_xy_equal_the_reference(gpu, name):
    """`Node.path_x` / `path_y` of every node (what the reference's draw_graph plots, rrt_04:1165-1167), against
    goldens made by running the reference (oracle/gen_golden_paths.py): the mirror classes rebuild the polylines on the
    host from the device's per-iteration trace -- extension edges (rrt_01 nodes; rrt_04 nodes appended when choose_parent
    returned None, :1066-1067), re-steered edges under a chosen parent (:1279) and rewired edges (:1359) -- bit for bit."""
    import random
    import rrt_amd
    g = np.load(util.GOLDEN + "/" + name + ".npz")
    kw = dict(start=list(g["kw_start"]), goal=list(g["kw_goal"]), obstacle_list=[tuple(o) for o in g["obstacles"]],
              rand_area=list(g["kw_rand_area"]), expand_dis=float(g["kw_expand_dis"]),
              path_resolution=float(g["kw_path_resolution"]), goal_sample_rate=int(g["kw_goal_sample_rate"]),
              max_iter=int(g["kw_max_iter"]), play_area=list(g["kw_play_area"]) if g["kw_play_area"].size else None,
              robot_radius=float(g["kw_robot_radius"]))
    random.seed(int(g["seed"]))
    if "rrt01" in name:
        rrt = rrt_amd.RRT(**kw)
    else:
        rrt = rrt_amd.RRTStar(sobol_sampler=False, connect_circle_dist=float(g["kw_connect_circle_dist"]),
                              search_until_max_iter=bool(g["kw_search_until_max_iter"]), **kw)
    path = rrt.planning(animation=False)
    assert len(rrt.node_list) == len(g["x"])
    assert (path is None) == (len(g["path"]) == 0) and (path is None or np.array_equal(np.array(path), g["path"]))
    off = 0
    for i, nd in enumerate(rrt.node_list):
        k = int(g["path_len"][i])
        assert nd.x == g["x"][i] and nd.y == g["y"][i], i
        assert list(nd.path_x) == list(g["path_x"][off:off + k]) and list(nd.path_y) == list(g["path_y"][off:off + k]), \
            "node %d (parent %d)" % (i, int(g["parent"][i]))
        off += k


def test_result_table_device_to_device_copy(gpu):
    """rrtx_copy_results_device: the 16-byte records the multi-GPU gather sends, copied device -> device into a torch
    tensor (no host round trip), equal the host table of rrtx_get_results."""
    import importlib
    import torch
    import rrt_amd
    sharding = importlib.import_module("robotics-path-planning_amd.sharding")
    A = rrt_amd._abi
    kw = util.c2_kwargs(600)
    seeds = list(range(1, 33))
    h = A.Handle(A.ALGO_RRT_STAR, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], kw["path_resolution"],
                 kw["goal_sample_rate"], kw["max_iter"], robot_radius=0.0, connect_circle_dist=50.0,
                 search_until_max_iter=True, n_instances=len(seeds))
    try:
        h.set_obstacles(kw["obstacles"])
        h.seed_instances(seeds)
        h.plan(strict=True)
        pc, nn, st = h.get_results()
        t = torch.zeros((len(seeds), 2), dtype=torch.int64, device="cuda:0")
        h.copy_results_device(t.data_ptr(), t.numel() * 8)
        pc2, nn2, st2 = sharding.unpack_records(t.cpu().numpy())
        assert np.array_equal(pc, pc2) and np.array_equal(nn, nn2) and np.array_equal(st, st2)
        with pytest.raises(A.RrtxError):
            h.copy_results_device(t.data_ptr(), 8)      # capacity too small
    finally:
        h.close()


def test_gpu_polyline_pool_overflow_is_replanned_with_a_larger_pool(gpu, monkeypatch):
    """rrt_05 / rrt_06 keep every edge polyline in a per-instance pool (edges replaced by rewire stay allocated).  An
    instance that runs out of pool is planned again with a pool four times as large, twice if need be, instead of
    failing the batch (round-1 VERDICT: capacity limits).  RRTX_POOL_POINTS_PER_NODE shrinks the pool for the test."""
    import oracle
    import rrt_amd
    g = util.load_golden(util.GOLDEN + "/rrt05_drv_s42_it500.npz")
    seeds = [42, 43, 44, 45]
    monkeypatch.setenv("RRTX_POOL_POINTS_PER_NODE", "4")
    out = util.run_gpu_dubins(g, seeds, max_iter=1500)       # plan(strict=True): raises if an instance stays overflowed
    for i, s in enumerate(seeds):
        r = oracle.plan_dubins(g["start"], g["goal"], g["obstacles"], g["rand_area"], 1500, seed=s)
        util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "seed %d" % s)
        assert np.array_equal(out["polys"][i][1], r["poly_x"]) and np.array_equal(out["yaws"][i], r["yaw"])
        assert (out["paths"][i] is None) == (r["path"] is None)
        if r["path"] is not None:
            assert np.array_equal(out["paths"][i], r["path"])
    monkeypatch.setenv("RRTX_NO_RETRY", "1")
    with pytest.raises(rrt_amd._abi.RrtxError):
        util.run_gpu_dubins(g, seeds, max_iter=1500)
    monkeypatch.delenv("RRTX_NO_RETRY")
    g6 = util.load_golden(util.GOLDEN + "/rrt06_drv_s42_it200.npz")
    monkeypatch.setenv("RRTX_POOL_POINTS_PER_NODE", "2")
    bp = rrt_amd.BatchPlanner("rrt_star_reeds_shepp", [42, 43], list(g6["start"]), list(g6["goal"]),
                              [tuple(o) for o in g6["obstacles"]], list(g6["rand_area"]), expand_dis=3.0,
                              goal_sample_rate=10, max_iter=200, robot_radius=0.6, search_until_max_iter=True,
                              curvature=2.0, step_size=0.1)
    try:
        bp.plan()
        assert not bp.partial
        assert np.array_equal(bp.path(0), g6["path"]) and np.array_equal(bp.yaw(0), g6["yaw"])
    finally:
        bp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["rrt_star", "informed", "bitstar"])
def test_batch_planner_sharded_over_handles_equals_one_handle(gpu, algo):
    """BatchPlanner(devices=[0, 0, 0]): three handles on device 0, planned concurrently by rrtx_plan_many (one native host
    thread per handle -- the single-process multi-GPU form) against ONE handle holding the whole batch: result
    table, trees, paths and RNG states equal bit for bit, instance by instance (the sharding changes no result)."""
    import random as _r
    import rrt_amd
    n = 25                                             # uneven split: 9 + 8 + 8
    seeds = list(range(101, 101 + n))
    kw = dict(seeds=seeds)
    if algo == "rrt_star":
        c2 = util.c2_kwargs(1500)
        kw.update(start=c2["start"], goal=c2["goal"], obstacle_list=c2["obstacles"], rand_area=c2["rand_area"],
                  expand_dis=2.0, path_resolution=0.25, goal_sample_rate=5, max_iter=1500, search_until_max_iter=True)
    elif algo == "informed":
        g = util.load_golden(util.GOLDEN + "/rrt07_drv_mt_s42_it2000.npz")
        k7 = util.informed_kwargs_from_golden(g)
        rr = _r.Random(3)
        kw.update(start=k7["start"], goal=k7["goal"], obstacle_list=k7["obstacles"], rand_area=k7["rand_area"],
                  expand_dis=0.5, goal_sample_rate=10, max_iter=600,
                  goals=[[k7["goal"][0] + rr.uniform(-0.5, 0.5), k7["goal"][1] + rr.uniform(-0.5, 0.5)] for _ in seeds])
    else:
        obst = [(5, 5, 0.5), (9, 6, 1), (7, 5, 1), (1, 5, 1), (3, 6, 1), (7, 9, 1)]
        rr = _r.Random(4)

        def free():
            while True:
                x, y = rr.uniform(-1, 14), rr.uniform(-1, 14)
                if all((x - ox) ** 2 + (y - oy) ** 2 > (r + 0.2) ** 2 for ox, oy, r in obst):
                    return [x, y]
        kw.update(start=[-1.0, 0.0], goal=[3.0, 8.0], obstacle_list=obst, rand_area=[-2.0, 15.0], max_iter=80,
                  starts=[free() for _ in seeds], goals=[free() for _ in seeds])
    one = rrt_amd.BatchPlanner(algo, **kw)
    many = rrt_amd.BatchPlanner(algo, devices=[0, 0, 0], **kw)
    try:
        assert [hi - lo for lo, hi in many.shards] == [9, 8, 8] and len(many.handles) == 3
        r1 = one.plan()
        r3 = many.plan()
        for a, b in zip(r1, r3):
            assert np.array_equal(a, b)
        for i in range(n):
            util.assert_tree_equal(many.tree(i), one.tree(i), "%s instance %d" % (algo, i))
            p1, p3 = one.path(i), many.path(i)
            assert (p1 is None) == (p3 is None) and (p1 is None or np.array_equal(p1, p3))
            assert many.rng_state(i) == one.rng_state(i)
        s1, s3 = one.stats(), many.stats()
        for k in ("iterations", "edges_unique", "edges_ref", "total_nodes", "rewires"):
            assert s1[k] == s3[k], k
        assert len(s3["per_shard"]) == 3
    finally:
        one.close()
        many.close()


def _orc_bench_tree(a):
    import oracle
    w, kw, it, sd = a
    if w == "c3":
        kc = dict(kw)
        kc.pop("algo")
        kc["max_iter"] = it
        r = oracle.plan_informed(seed=sd, exact_pow=False, **kc)
        return r["x"], r["y"], r["cost"], r["parent"], r["path"], None
    r = oracle.plan_dubins(kw["start"], kw["goal"], kw["obstacles"], kw["rand_area"], it, seed=sd)
    return r["x"], r["y"], r["cost"], r["parent"], r["path"], r["yaw"]


@pytest.mark.gpu
@pytest.mark.parametrize("w", ["c3", "c5"])
def test_gpu_bench_size_batches_sampled_against_oracle(gpu, w):
    """The C3 (rrt_07, 1 024 x 20 000) and C5 (rrt_05, 1 536 x 5 000) batches exactly as bench.py builds and plans them
    (its own Workload class: same map, constants, seeds, instance count, one handle), eight trees spread over the batch
    against the oracle at the bench's size, bit for bit (round 2 held this in tools/ only)."""
    import concurrent.futures as cf
    import importlib
    import types
    import bench
    import rrt_amd
    sharding = importlib.import_module("robotics-path-planning_amd.sharding")
    a = types.SimpleNamespace(workload=w, instances=None, max_iter=None, obstacles=None, warmup_max_iter=0)
    wl = bench.Workload(a, np, util, rrt_amd)
    wl.prepare(0, sharding)
    assert (wl.B, wl.max_iter) == {"c3": (1024, 20000), "c5": (1536, 5000)}[w]
    pick = sorted(set(int(v) for v in np.linspace(0, wl.B - 1, 8)))
    with cf.ProcessPoolExecutor(max_workers=8) as ex:
        fut = [ex.submit(_orc_bench_tree, (w, wl.kw, wl.max_iter, wl.seeds[i])) for i in pick]
        h = wl.make_handle(wl.max_iter, 0)
        try:
            wl.step(h)
            st = h.get_stats()
            assert st["iterations"] == wl.B * wl.max_iter
            for i, f in zip(pick, fut):
                ox, oy, oc, op, opath, oyaw = f.result()
                util.assert_tree_equal(h.get_tree(i), (ox, oy, oc, op), "%s instance %d" % (w, i))
                path = h.get_path(i)
                assert (path is None) == (opath is None)
                if path is not None:
                    assert np.array_equal(np.asarray(path)[:, :2], np.asarray(opath)[:, :2])
                if oyaw is not None:
                    assert np.array_equal(h.get_yaw(i), oyaw)
            if w == "c3":
                assert st["q16_fallbacks"] > 0      # the one-pass 16-bit first stage ran (and handed some queries down)
        finally:
            h.close()


@pytest.mark.gpu
def test_gpu_informed_near_set_overflow_is_replanned_on_the_large_shape(gpu):
    """rrt_07's near radius 50*sqrt(ln n / n) is not capped (:1139): on a 5 x 5 area it covers the whole tree, so the near
    set outgrows the 512 LDS candidate slots of the product shape after ~520 nodes.  rrtx_plan then plans the instance again
    on the 2 048-slot instantiation (one workgroup per CU, ~124 KB of LDS; round-2 ADVICE: that path had no test): tree,
    path and RNG state equal the oracle's, rrtx_stats.replanned counts it.  Beyond 2 048 candidates the instance ends with
    RRTX_ST_OVERFLOW and the call returns RRTX_PARTIAL -- for that instance only: an instance whose start lies inside an
    obstacle (every extension collides, the tree stays at its root) completes in the same batch."""
    import oracle
    import rrt_amd
    A = rrt_amd._abi
    obst = [(2.5, 2.5, 0.3), (1.0, 3.5, 0.25), (3.8, 1.2, 0.25)]
    kw = dict(start=[0.5, 0.5], goal=[4.5, 4.5], obstacles=obst, rand_area=[0.0, 5.0], expand_dis=0.08,
              goal_sample_rate=10, max_iter=1500, sobol=0)
    seeds = [3, 4]
    out = util.run_gpu_informed(kw, seeds)
    assert out["stats"]["replanned"] == 2 and out["stats"]["near_unique"] > 512 * 500
    for i, s in enumerate(seeds):
        r = oracle.plan_informed(seed=s, **kw)
        assert len(r["x"]) > 900
        util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "seed %d" % s)
        assert (out["paths"][i] is None) == (r["path"] is None)
        if r["path"] is not None:
            assert np.array_equal(out["paths"][i], r["path"]) and out["results"][0][i] == r["c_best"]
        assert out["rng"][i][1][624] == r["rng"].pos and out["rng"][i][1][0] == r["rng"].mt[0]
    # more than 2 048 candidates: per-instance overflow, the walled-in instance of the same batch completes
    c_min, c = rrt_amd.informed_rotation(kw["start"], kw["goal"])
    h = A.Handle(A.ALGO_INFORMED, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], 1.0, 10, 3200,
                 n_instances=2, informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
    try:
        h.set_obstacles(obst)
        h.seed_instances([3, 4])
        h.set_instance(1, [2.5, 2.5], [4.5, 4.5])
        cm1, c1 = rrt_amd.informed_rotation([2.5, 2.5], [4.5, 4.5])
        h.set_instance_rotation(1, [c1[0, 0], c1[0, 1], c1[1, 0], c1[1, 1]], cm1)
        assert h.plan() == A.RRTX_PARTIAL and "RRTX_ST_OVERFLOW" in h.last_error()
        pc, nn, st = h.get_results()
        assert st[0] & A.ST_OVERFLOW and not (st[1] & A.ST_FAILED) and st[1] & A.ST_DONE
        assert nn[1] == 1 and nn[0] > 2048
    finally:
        h.close()


@pytest.mark.gpu
def test_gpu_informed_nodes_leaving_the_16bit_grid_fall_back(gpu, monkeypatch):
    """rrt_07 kernel: the one-pass 16-bit first stage holds while every node lies on the grid square (sampling square plus
    a margin).  RRTX_Q16_PAD=-0.02 shrinks the square INTO the sampling area, so samples near the border are off the grid
    (their nearest query takes the f32 pass) and the first node that steps off it switches the instance to the f32 / f64
    passes for good; RRTX_Q16=0 never uses the grid.  All three give the oracle's trees."""
    import oracle
    g = util.load_golden(util.GOLDEN + "/rrt07_c3_sobol_s1_it3000.npz")
    kw = util.informed_kwargs_from_golden(g)
    kw["max_iter"] = 2500
    kw["start"], kw["goal"] = [4.0, 4.0], [96.0, 96.0]      # inside the shrunk grid; the tree is free to walk to the border
    kw["obstacles"] = [o for o in kw["obstacles"] if (o[0] - 4) ** 2 + (o[1] - 4) ** 2 > (o[2] + 1) ** 2
                       and (o[0] - 96) ** 2 + (o[1] - 96) ** 2 > (o[2] + 1) ** 2]
    seeds = list(range(21, 29))
    ref = [oracle.plan_informed(seed=s, **kw) for s in seeds]
    for env in ({}, {"RRTX_Q16_PAD": "-0.02"}, {"RRTX_Q16": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = util.run_gpu_informed(kw, seeds)
        for k in env:
            monkeypatch.delenv(k)
        for i, s in enumerate(seeds):
            r = ref[i]
            util.assert_tree_equal(out["trees"][i], (r["x"], r["y"], r["cost"], r["parent"]), "%s seed %d" % (env, s))
            assert (out["paths"][i] is None) == (r["path"] is None)
        if env.get("RRTX_Q16") == "0":
            assert out["stats"]["q16_fallbacks"] == 0


def _c4_instances(lo, hi):
    """Instances lo..hi-1 of the C4 generator (bench.py / SURVEY 8d): start / goal from random.Random(2000 + i) outside
    the obstacles, planner seed 1000 + i."""
    import random
    obst = [(5, 5, 0.5), (9, 6, 1), (7, 5, 1), (1, 5, 1), (3, 6, 1), (7, 9, 1)]

    def free_point(rng):
        while True:
            x, y = rng.uniform(-1, 14), rng.uniform(-1, 14)
            if all((x - ox) ** 2 + (y - oy) ** 2 > r ** 2 for ox, oy, r in obst):
                return [x, y]
    starts, goals, seeds = [], [], []
    for i in range(lo, hi):
        rng = random.Random(2000 + i)
        starts.append(free_point(rng))
        goals.append(free_point(rng))
        seeds.append(1000 + i)
    return obst, starts, goals, seeds


@pytest.mark.gpu
def test_gpu_bitstar_bounded_launches_carry_instances_over(gpu, monkeypatch):
    """BIT* launches are bounded and run persistent waves over a device-side work queue.  With
    RRTX_BITSTAR_TRIPS=37 no instance finishes in one launch: each stores its whole state (LDS columns, queues, RNG) when
    its trips are used up, is queued again, and another wave resumes it -- dozens of times.  RRTX_BITSTAR_GRID=5 makes five
    persistent waves pull all 96 instances from the queue.  Trees, paths, popped-edge traces and RNG states equal the
    unbounded single-launch run's and the reference goldens."""
    obst, starts, goals, seeds = _c4_instances(0, 96)
    ref = util.run_gpu_bitstar(obst, [-2.0, 15.0], 80, seeds, starts, goals, trace_instance=7)
    assert ref["stats"]["launches"] == 1
    for env in ({"RRTX_BITSTAR_TRIPS": "37"}, {"RRTX_BITSTAR_GRID": "5"}, {"RRTX_BITSTAR_TRIPS": "11", "RRTX_BITSTAR_GRID": "3"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = util.run_gpu_bitstar(obst, [-2.0, 15.0], 80, seeds, starts, goals, trace_instance=7)
        for k in env:
            monkeypatch.delenv(k)
        if "RRTX_BITSTAR_TRIPS" in env:
            assert out["stats"]["launches"] > 10
        for a, b in zip(out["results"], ref["results"]):
            assert np.array_equal(a, b), env
        for i in range(len(seeds)):
            util.assert_tree_equal(out["trees"][i], ref["trees"][i], "%s instance %d" % (env, i))
            p, q = out["paths"][i], ref["paths"][i]
            assert (p is None) == (q is None) and (p is None or np.array_equal(p, q))
            assert out["rng"][i] == ref["rng"][i]
        assert np.array_equal(out["trace"][0], ref["trace"][0]) and np.array_equal(out["trace"][1], ref["trace"][1])
        assert out["stats"]["edges_unique"] == ref["stats"]["edges_unique"]
    monkeypatch.setenv("RRTX_BITSTAR_TRIPS", "23")
    for path in util.golden_files("rrt08")[:4]:
        g = util.load_golden(path)
        out = util.run_gpu_bitstar([tuple(float(v) for v in o) for o in g["obstacles"]], [float(v) for v in g["rand_area"]],
                                   int(g["max_iter"]), [int(g["seed"])], [[float(v) for v in g["start"]]],
                                   [[float(v) for v in g["goal"]]], trace_instance=0)
        assert np.array_equal(out["trace"][0], g["tr_e0"]) and np.array_equal(out["trace"][1], g["tr_e1"])
        gx, gy = _bit_coords(g["vertex_ids"])
        x, y, cost, parent = out["trees"][0]
        assert np.array_equal(x, gx) and np.array_equal(y, gy) and np.array_equal(cost, g["g_scores"])
        assert out["rng"][0][1][624] == int(g["rng_pos_after"]) and out["rng"][0][1][0] == int(g["rng_word0_after"])


@pytest.mark.gpu
def test_gpu_bitstar_walled_in_instance_does_not_hold_the_batch(gpu):
    """4 096 C4 instances, one of them with its start inside an obstacle (the reference never returns from it; round 2:
    18 s on one wave = the time of the whole batch).  The whole batch now plans in well under half a second: the straggler
    ends RRTX_ST_OVERFLOW | RRTX_ST_REF_HANGS, every other instance completes and the batch equals the run without it."""
    import time
    import rrt_amd
    A = rrt_amd._abi
    obst, starts, goals, seeds = _c4_instances(0, 4096)
    bad = 1234
    starts_b = [list(v) for v in starts]
    starts_b[bad] = [7.0, 5.0]                      # inside the circle (7, 5, 1)
    bp = rrt_amd.BatchPlanner("bitstar", seeds, starts[0], goals[0], obst, [-2.0, 15.0], max_iter=80, starts=starts_b,
                              goals=goals)
    ok = rrt_amd.BatchPlanner("bitstar", seeds, starts[0], goals[0], obst, [-2.0, 15.0], max_iter=80, starts=starts,
                              goals=goals)
    try:
        bp.plan()                                    # first call pages the code object in
        bp.h.seed_instances(seeds)
        t0 = time.perf_counter()
        pc, nn, st = bp.plan()
        dt = time.perf_counter() - t0
        assert dt < 0.5, dt
        assert bp.partial and st[bad] & A.ST_OVERFLOW and st[bad] & A.ST_REF_HANGS
        assert [i for i, _ in bp.failed()] == [bad]
        pc0, nn0, st0 = ok.plan()
        keep = np.arange(4096) != bad
        assert np.array_equal(pc[keep], pc0[keep]) and np.array_equal(nn[keep], nn0[keep]) and np.array_equal(st[keep], st0[keep])
        assert (st[keep] & A.ST_DONE).all()
    finally:
        bp.close()
        ok.close()


@pytest.mark.gpu
def test_gpu_plan_in_bounded_steps(gpu):
    """rrtx_plan_begin / rrtx_plan_step: a plan as a sequence of bounded launches (rrtx_set_launch_bound).  Between steps
    the result records of finished instances are final; the completed plan equals rrtx_plan's.  rrt_04 iteration kernel +
    final goal search, rrt_07, and BIT* (where run lengths differ from instance to instance, so the pending count falls step
    by step)."""
    import rrt_amd
    A = rrt_amd._abi
    # BIT*
    obst, starts, goals, seeds = _c4_instances(0, 64)
    ref = util.run_gpu_bitstar(obst, [-2.0, 15.0], 80, seeds, starts, goals)
    c_min, c = rrt_amd.bitstar_rotation(starts[0], goals[0])
    h = A.Handle(A.ALGO_BITSTAR, starts[0], goals[0], [-2.0, 15.0], 2.0, 1.0, 0, 80, n_instances=64,
                 informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
    try:
        h.set_obstacles(obst)
        h.seed_instances(seeds)
        for i in range(64):
            cm, ci = rrt_amd.bitstar_rotation(starts[i], goals[i])
            h.set_instance(i, starts[i], goals[i])
            h.set_instance_rotation(i, [ci[0, 0], ci[0, 1], ci[1, 0], ci[1, 1]], cm)
        h.set_launch_bound(150)
        h.plan_begin()
        pend, done_at = [], {}
        while True:
            rc, n = h.plan_step()
            pend.append(n)
            st = h.get_results()[2]
            for i in np.nonzero(st & A.ST_DONE)[0]:
                done_at.setdefault(int(i), (len(pend), h.get_results()[0][i]))
            if n == 0:
                break
        assert len(pend) > 3 and pend == sorted(pend, reverse=True) and pend[-1] == 0
        assert any(0 < n < 64 for n in pend)         # instances finish in different launches
        pc, nn, st = h.get_results()
        for a, b in zip((pc, nn, st), ref["results"]):
            assert np.array_equal(a, b)
        for i, (_, cost_then) in done_at.items():
            assert cost_then == pc[i] or (np.isinf(cost_then) and np.isinf(pc[i]))     # final when first seen DONE
        for i in range(64):
            util.assert_tree_equal(h.get_tree(i), ref["trees"][i], "BIT* instance %d" % i)
    finally:
        h.close()
    # rrt_04 (iteration kernel in 500-iteration launches, then the goal search) and rrt_07 (400-iteration launches)
    g = util.load_golden(util.GOLDEN + "/rrt04_c2_s1_it4000.npz")
    kw = util.kwargs_from_golden(g)
    h = A.Handle(A.ALGO_RRT_STAR, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], kw["path_resolution"],
                 kw["goal_sample_rate"], kw["max_iter"], robot_radius=kw["robot_radius"],
                 connect_circle_dist=kw["connect_circle_dist"], search_until_max_iter=True, n_instances=1)
    try:
        h.set_obstacles(kw["obstacles"])
        h.seed_instances([int(g["seed"])])
        h.set_launch_bound(500)
        h.plan_begin()
        steps = 0
        while True:
            rc, n = h.plan_step()
            steps += 1
            if n == 0:
                break
        assert steps >= 9 and rc == 0
        util.assert_tree_equal(h.get_tree(0), (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
        assert np.array_equal(h.get_path(0), g["path"])
    finally:
        h.close()
    g = util.load_golden(util.GOLDEN + "/rrt07_c3_sobol_s1_it3000.npz")
    kw = util.informed_kwargs_from_golden(g)
    c_min, c = rrt_amd.informed_rotation(kw["start"], kw["goal"])
    h = A.Handle(A.ALGO_INFORMED, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], 1.0, kw["goal_sample_rate"],
                 kw["max_iter"], sampler=A.SAMPLER_SOBOL, n_instances=1,
                 informed_rot=[c[0, 0], c[0, 1], c[1, 0], c[1, 1]], informed_c_min=c_min)
    try:
        h.set_obstacles(kw["obstacles"])
        h.seed_instances([int(g["seed"])])
        h.set_launch_bound(400)
        h.plan_begin()
        steps = 0
        while h.plan_step()[1]:
            steps += 1
        assert steps >= 7
        util.assert_tree_equal(h.get_tree(0), (g["x"], g["y"], g["cost"], g["parent"]), g["name"])
    finally:
        h.close()



@pytest.mark.gpu
def test_native_rccl_gather_of_the_result_table(gpu):
    """rrtx_rccl_unique_id / rrtx_rccl_init / rrtx_rccl_gather_results: the path's one collective as a native ncclAllGather
    (RCCL opened with dlopen by librrtx.so, no framework in between).  The builder's box has one GPU: world size 1 -- the
    gathered table must be the handle's own result table; and bench.py's opt-in use of it (RRTX_BENCH_NATIVE_RCCL=1 with
    a forced world-size-1 process group) prints the same line as the torch.distributed gather."""
    import json
    import os
    import subprocess
    import sys
    import rrt_amd
    A = rrt_amd._abi
    kw = util.c2_kwargs(1200)
    h = A.Handle(A.ALGO_RRT_STAR, kw["start"], kw["goal"], kw["rand_area"], kw["expand_dis"], kw["path_resolution"],
                 kw["goal_sample_rate"], kw["max_iter"], robot_radius=0.0, connect_circle_dist=50.0,
                 search_until_max_iter=True, n_instances=24)
    try:
        h.set_obstacles(kw["obstacles"])
        h.seed_instances(list(range(1, 25)))
        uid = A.rccl_unique_id()
        assert len(uid) == 128
        h.rccl_init(uid, 0, 1)
        h.plan()
        pc, nn, st = h.get_results()
        gpc, gnn, gst = h.rccl_gather_results()
        assert np.array_equal(pc, gpc) and np.array_equal(nn, gnn) and np.array_equal(st, gst)
    finally:
        h.close()
    env = dict(os.environ, RRTX_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    lines = []
    for native in ("", "1"):
        e = dict(env)
        if native:
            e["RRTX_BENCH_NATIVE_RCCL"] = "1"
        r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--instances", "48", "--max-iter", "1500",
                            "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=e,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        lines.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    assert lines[1]["result_gather"].startswith("rrtx_rccl_gather_results") and lines[0]["result_gather"].startswith("torch")
    for k in ("paths_found", "final_path_cost_mean", "final_path_cost_min", "mean_nodes_per_tree", "instances_total"):
        assert lines[0][k] == lines[1][k], k
