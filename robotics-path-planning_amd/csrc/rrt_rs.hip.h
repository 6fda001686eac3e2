// rrt_rs.hip.h -- RRT*-Reeds-Shepp iteration kernel (gfx950).
// Reference: /root/reference/src_path_planning/10_path_planning_01_rrt_06_rrt_star_reeds_shepp_path.py (rrt_06)
//   planning :1530-1570, try_goal_path :1572-1582, steer :1584-1604 -> reeds_shepp_path_planning :1426-1441,
//   get_random_node :1658-1666 (three uniform draws, no goal sampling), get_nearest_node_index :1728-1734,
//   check_collision :1748-1762, find_near_nodes :1839-1863, choose_parent :1772-1813, rewire :1865-1899 with the
//   EFFECTIVE (last-defined) Euclidean calc_new_cost :1901-1903 and propagate_cost_to_leaves :1905-1911,
//   search_best_goal_node :1815-1836.
//
// One wavefront (64 lanes) per planning instance; many instances per CU.  The tree of one rrt_06 run is small
// (at most 2 * max_iter + 1 nodes: try_goal_path can add a second node per iteration) and every step of the
// reference depends on the one before, so the parallelism inside an instance is in the Reeds-Shepp steer:
//   * the 12 word families x 4 symmetries of generate_path (:1286-1342) are 48 independent closed-form
//     evaluations: one lane each (rpp::rs_variant);
//   * set_path's order-dependent filtering (:1061-1080) is settled front to back in 48 wave-uniform steps, the
//     arg-min of :1436 is a wave reduction, and the course is laid out one lane per segment (segment yaws are a
//     running sum, so each lane knows its own origin yaw; the origins' positions are then a 5-step chain);
//   * the course's points are independent once the segment origins are known: all lanes generate them
//     (rpp::rs_point), test them against the LDS obstacle tile as they are produced (:1748-1762) and write them
//     behind the instance's polyline pool, where they stay if the edge is kept.
// choose_parent / rewire steer one cooperative edge after the other, and only the edges that can change the result
// (da.eager = 0, the default): a candidate's cost does not depend on its steer, so choose_parent tries candidates in
// ascending (cost, list position) order until one is feasible, and rewire walks near_inds in the reference's order but
// steers an entry only when improved_cost (:1888) holds for the node's current pose and cost -- repeated entries of
// near_inds (`dist_list.index`, :1861) and nodes moved by an earlier rewire are seen exactly as the reference sees
// them.  da.eager = 1 (RRTX_RS_EAGER=1) steers every candidate like the reference; same trees either way (tests).
// Scans use correctly rounded squares as a filter and the reference's pow-based distance for near-ties and hits.
// Cost propagation is a level-synchronous sweep over the parent array.
#pragma once
#include "rpp_rs.h"
#include "rrt_dubins.hip.h"

namespace rppr {

using rppk::Ctx;
using rppk::Inst;
using rppd::DubArgs;

constexpr int TPB = 64;
constexpr int MAX_OBS = 64;   // obstacle tile in LDS (rrtx_plan refuses larger obstacle sets for this planner)
constexpr int RS_ST_RAISES = 32;   // include/rrtx.h RRTX_ST_REF_RAISES

#ifdef RRTX_PHASE_TIMERS
#define RS_T0 int64_t rt_ = (int64_t)__builtin_amdgcn_s_memtime();
#define RS_T(k) do { if (threadIdx.x == 0) { int64_t t_ = (int64_t)__builtin_amdgcn_s_memtime(); sh.ph[k] += t_ - rt_; rt_ = t_; } } while (0)
#define RS_M0 int64_t mt_ = (int64_t)__builtin_amdgcn_s_memtime();
#define RS_M(k) do { if (threadIdx.x == 0) { int64_t t_ = (int64_t)__builtin_amdgcn_s_memtime(); sh.ph[k] += t_ - mt_; mt_ = t_; } } while (0)
#else
#define RS_T0
#define RS_T(k) do { } while (0)
#define RS_M0
#define RS_M(k) do { } while (0)
#endif

struct ShR {
#ifdef RRTX_PHASE_TIMERS
  int64_t ph[16];
#endif
  rpp::MT rng;
  double ox[MAX_OBS], oy[MAX_OBS], othr[MAX_OBS];
  double vd[48][5];
  char vct[48][8];
  int32_t vn[48];
  double sdx[5], sdy[5];      // displacement of each segment's last point from its origin
  rpp::RsCourse course;
  double cost_len;           // sum(|course_lengths|) of the chosen path (:1600)
  double ex, ey, eyaw;       // last point of the course = pose of the node steer() returns (:1592-1594)
  double rx, ry, ryaw;
  int32_t sel, any_hit, flag;
};

// first minimum of (v, idx) over the wave: lowest v, lowest idx among equals
__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
  for (int o = 32; o >= 1; o >>= 1) {
    const double ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(idx, o);
    if (ov < v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
}

// Cooperative steer (fx,fy,fyaw) -> (tx,ty,tyaw).  Wave-uniform return: 1 a node exists (course in sh.course, its
// points written at px/py/pyaw when they fit `room`, *coll = check_collision fails, *npts = points), 0 steer
// returns None, < 0 the reference raises (-3 ZeroDivisionError, -4 ValueError), -1 pool full.
__device__ __noinline__ int rs_edge(const DubArgs& da, int m, ShR& sh, double fx, double fy, double fyaw, double tx,
                                    double ty, double tyaw, double* __restrict__ px, double* __restrict__ py,
                                    double* __restrict__ pyaw, int64_t room, int* coll, int* npts) {
  const int lane = threadIdx.x;
  RS_T0
  // ---- generate_path :1286-1342: one (word family, symmetry) per lane
  int st = 0, nn = 0;
  double L = 0.0;
  uint32_t code = 0;
  rpp::RsFrame F;
  F.c = F.s = 0.0;
  // the four libm calls every variant starts with -- cos | sin of the start yaw (the frame, :1286-1296), sin | cos of the
  // yaw difference (the words' phi) -- do not depend on each other: four lanes, one call each, at the same time
  double t4 = 0.0;
  if (lane < 4) {
    const double arg = lane < 2 ? fyaw : tyaw - fyaw;
    t4 = (lane == 0 || lane == 3) ? rpp_glibc_cos(arg) : rpp_glibc_sin(arg);
  }
  const double c_sy = __shfl(t4, 0), s_sy = __shfl(t4, 1), s_dth = __shfl(t4, 2), c_dth = __shfl(t4, 3);
  if (lane < 48) {
    rpp::rs_frame_sc(fx, fy, fyaw, tx, ty, tyaw, da.curvature, da.step_size, c_sy, s_sy, &F);
    st = rpp::rs_variant_sc(lane >> 2, lane & 3, F, s_dth, c_dth, sh.vd[lane], sh.vct[lane], &nn);
    if (st == 1) {
      L = rpp::rs_sum_abs(sh.vd[lane], nn);
      code = rpp::rs_code(sh.vct[lane]);
    }
    sh.vn[lane] = nn;
  }
  RS_T(0);
  // ---- the reference walks the variants in order and leaves at the first one that raises or reports
  // "Step size too large" (:1071-1074); nothing before it matters then
  const unsigned long long term = __ballot(st < 0 || st == 2);
  if (term) {
    const int sk = __shfl(st, __ffsll((long long)term) - 1);
    __syncthreads();
    return sk < 0 ? sk : 0;
  }
  // ---- set_path :1061-1080: variant k is kept unless an EARLIER KEPT one has the same letters and is not longer
  // by more than step; order-dependent, so the kept ones are settled front to back (48 uniform steps)
  const double step = da.step_size * da.curvature;
  const bool cand = (st == 1) && !(L <= step);
  bool skip = false;
  for (int i = -1;;) {
    const unsigned long long km = __ballot(cand && !skip) & (i < 0 ? ~0ULL : ~((2ULL << i) - 1ULL));   // still kept, behind i
    if (!km) break;
    i = __ffsll((long long)km) - 1;
    const double Li = __shfl(L, i);
    const uint32_t ci = __shfl(code, i);
    if (lane > i && st == 1 && code == ci && (Li - L) <= step) skip = true;
  }
  // paths.index(min(paths, key=abs(L))) :1436 -- first minimum
  const bool kept = cand && !skip;
  double bv = kept ? rpp::dabs(L / da.curvature) : rpp::dinf();
  int sel = kept ? lane : 0x7fffffff;
  wave_argmin(bv, sel);
  RS_T(1);
  __syncthreads();
  if (sel == 0x7fffffff) return 0;   // no path: steer returns None
  // ---- generate_local_course :1355-1377 laid out for random access: one lane per segment, then a short chain
  const int nl = sh.vn[sel];
  rpp::RsCourse& C = sh.course;
  // four lanes per segment: cos | sin of its origin yaw, sin | cos of its length -- the four calls of rs_seg_trig /
  // rs_delta at once instead of one after the other on the segment's lane; lane 4 * s then lays segment s out
  {
    const int sg = lane >> 2, wh = lane & 3;
    double oyaw = 0.0, len = 0.0, tv = 0.0;
    char md = 'S';
    if (sg < nl) {
      for (int q = 0; q < sg; q++) oyaw = rpp::rs_yaw_after(sh.vd[sel][q], sh.vct[sel][q], oyaw);
      len = sh.vd[sel][sg];
      md = sh.vct[sel][sg];
      const double arg = wh < 2 ? oyaw : len;
      tv = (wh == 0 || wh == 3) ? rpp_glibc_cos(arg) : rpp_glibc_sin(arg);
    }
    const int b4 = lane & ~3;
    const double c0 = __shfl(tv, b4), s0 = __shfl(tv, b4 + 1), sd = __shfl(tv, b4 + 2), cd = __shfl(tv, b4 + 3);
    if (sg < nl && wh == 0) {
      const double ds = da.step_size * da.curvature;
      const double d_dist = len >= 0.0 ? ds : -ds;
      const double q = (len - 0.0) / d_dist;            // np.arange(0.0, length, d_dist)
      const long cnt = (q > 0.0) ? (long)__builtin_ceil(q) : 0;
      const double cs = c0, sn = (md == 'S') ? s0 : -s0;   // rs_seg_trig
      double dx, dy;
      rpp::rs_delta_sc(len, md, da.curvature, cs, sn, sd, cd, &dx, &dy);
      C.len[sg] = len;
      C.ddist[sg] = d_dist;
      C.ct[sg] = md;
      C.oyaw[sg] = oyaw;
      C.cs[sg] = cs;
      C.sn[sg] = sn;
      C.cnt[sg] = (int32_t)cnt;
      sh.sdx[sg] = dx;
      sh.sdy[sg] = dy;
    }
  }
  if (lane == 32) {
    C.cg = F.c;     // cos(-syaw), sin(-syaw) of :1411-1417 from the frame's cos / sin(syaw) (even / odd bit for bit)
    C.sg = -F.s;
    C.sx = fx;
    C.sy = fy;
    C.syaw = fyaw;
    C.maxc = da.curvature;
    double s = 0.0;
    for (int i = 0; i < nl; i++) s += rpp::dabs(sh.vd[sel][i] / da.curvature);
    sh.cost_len = s;
  }
  __syncthreads();
  if (lane == 0) {
    double ox = 0.0, oy = 0.0;
    int tot = 0;
    for (int sgm = 0; sgm < nl; sgm++) {
      C.ox[sgm] = ox;
      C.oy[sgm] = oy;
      C.first[sgm] = tot;
      tot += C.cnt[sgm] + 1;
      ox = ox + sh.sdx[sgm];
      oy = oy + sh.sdy[sgm];
    }
    C.ct[nl] = 0;
    C.first[nl] = tot;
    C.total = tot;
    C.nl = nl;
    sh.any_hit = 0;
  }
  __syncthreads();
  RS_T(2);
  const int total = C.total;
  *npts = total;
  if (total <= 0) return 0;   // `if not px` :1588
  if (total > room) return -1;
  int hit = 0;
  for (int k = lane; k < total; k += TPB) {
    double wx, wy, wyaw;
    rpp::rs_point(C, k, &wx, &wy, &wyaw);
    px[k] = wx;
    py[k] = wy;
    pyaw[k] = wyaw;
    for (int o = 0; o < m; o++) {
      const double dx = sh.ox[o] - wx, dy = sh.oy[o] - wy;
      if (dx * dx + dy * dy <= sh.othr[o]) hit = 1;
    }
    if (k == total - 1) {
      sh.ex = wx;
      sh.ey = wy;
      sh.eyaw = wyaw;
    }
  }
  if (hit) sh.any_hit = 1;
  __syncthreads();
  RS_T(3);
  *coll = sh.any_hit;
  return 1;
}

#ifndef RS_WAVES
#define RS_WAVES 4   // measured: 2 waves/SIMD (no spills) 3.3, 4 waves 5.8, 6 waves 6.5 plans/ms on the c6 workload
#endif
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(RS_WAVES, RS_WAVES))) void rrt_rs_kernel(Ctx c, DubArgs da, int iters) {
  __shared__ ShR sh;
  const int inst = c.inst_map ? c.inst_map[blockIdx.x] : blockIdx.x;
  const int lane = threadIdx.x;
  Inst* I = c.inst + inst;
  if (I->status & 1) return;
  const int64_t off = (int64_t)inst * c.stride;
  double* __restrict__ x = c.x + off;
  double* __restrict__ y = c.y + off;
  double* __restrict__ yaw = da.yaw + off;
  double* __restrict__ cost = c.cost + off;
  int32_t* parent = c.parent + off;
  int32_t* mark = c.hits + off;       // propagation levels
  int32_t* near = c.stack + off;      // near_inds of the current iteration
  double* ndist = da.dscr + off;      // their distances (for the `.index()` collapse)
  int64_t* poff = da.poff + off;
  int32_t* plen = da.plen + off;
  const int64_t pslot = da.pool_slot ? da.pool_slot[inst] : inst;
  double* pool_x = da.pool_x + pslot * da.pool_cap;
  double* pool_y = da.pool_y + pslot * da.pool_cap;
  double* pool_w = da.pool_yaw + pslot * da.pool_cap;
  const int m = c.m;

  for (int i = lane; i < 624; i += TPB) sh.rng.mt[i] = I->rng.mt[i];
  for (int i = lane; i < m; i += TPB) {
    sh.ox[i] = c.ox[i];
    sh.oy[i] = c.oy[i];
    sh.othr[i] = c.othr[i];
  }
  if (lane == 0) sh.rng.pos = I->rng.pos;
#ifdef RRTX_PHASE_TIMERS
  if (lane < 16) sh.ph[lane] = 0;
  const int64_t tk0_ = (int64_t)__builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  int n = I->n, it = I->it;
  int64_t pool_used = da.pool_used[inst];
  const double gx = I->goal[0], gy = I->goal[1], gyaw = I->goal[2];
  if (it == 0 && lane == 0) {
    yaw[0] = I->start[2];
    poff[0] = 0;
    plen[0] = 0;
  }
  __syncthreads();
  int64_t s_iter = 0, s_e = 0, s_nh = 0, s_rw = 0, s_pr = 0, s_sn = 0, s_pts = 0, s_ref = 0;
  int stop = 0, done_early = 0, raised = 0;

  // search_best_goal_node :1815-1836: lowest cost inside both thresholds, first index among equals
  auto goal_search = [&](double& gb, int& gi) {
    double best = rpp::dinf();
    int bidx = 0x7fffffff;
    for (int i = lane; i < n; i += TPB) {
      if (rpp::py_hypot(x[i] - gx, y[i] - gy) <= da.goal_xy_th && rpp::dabs(yaw[i] - gyaw) <= da.goal_yaw_th) {
        const double cc = cost[i];
        if (cc < best) {
          best = cc;
          bidx = i;
        }
      }
    }
    wave_argmin(best, bidx);
    gb = best;
    gi = bidx;
  };
  // steer result -> stop flags; returns true when the iteration has to end
  auto fatal = [&](int rc) {
    if (rc == -1) {
      stop = 1;           // polyline pool full
      return true;
    }
    if (rc < -1) {
      stop = 1;
      raised = rc;        // the reference raises inside reeds_shepp_path_planning
      return true;
    }
    return false;
  };
  // append a node whose edge polyline (np points) has just been written behind the pool
  auto append_node = [&](int par, double cst, int np) {
    if (lane == 0) {
      x[n] = sh.ex;
      y[n] = sh.ey;
      yaw[n] = sh.eyaw;
      cost[n] = cst;
      parent[n] = par;
      poff[n] = pool_used;
      plen[n] = np;
    }
    pool_used += np;
    n++;
    __syncthreads();
  };

  RS_M0
  for (int step = 0; step < iters && it < c.max_iter && !stop; step++, it++) {
    s_iter++;
    RS_M(13);
    // ---------------- get_random_node :1658-1666
    if (lane == 0) {
      sh.rx = rpp::mt_uniform(&sh.rng, c.rand_min, c.rand_max);
      sh.ry = rpp::mt_uniform(&sh.rng, c.rand_min, c.rand_max);
      sh.ryaw = rpp::mt_uniform(&sh.rng, -rpp::kPi, rpp::kPi);
    }
    __syncthreads();
    const double rx = sh.rx, ry = sh.ry, ryaw = sh.ryaw;
    RS_M(4);
    // ---------------- get_nearest_node_index :1728-1734 (x, y only; first minimum)
    // The reference's distance is dx**2 + dy**2 with libm pow (rpp::py_d2); correctly rounded squares
    // (rpp::fast_d2) are within 2^-51 relative of it, so they decide everything except near-ties: the exact form is
    // evaluated only for nodes within FILTER_EPS of the fast minimum.
    // (scans: four nodes per lane and round trip -- the coordinates of 256 nodes are requested together; a loop that waits
    // for every pair of loads spends a memory round trip per 64 nodes)
    double fb = rpp::dinf();
    for (int i0 = lane; i0 < n; i0 += 4 * TPB) {
      double vx[4], vy[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * TPB;
        const int ii = i < n ? i : i0;
        vx[u] = x[ii];
        vy[u] = y[ii];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double f = (i0 + u * TPB < n) ? rpp::fast_d2(vx[u] - rx, vy[u] - ry) : rpp::dinf();
        fb = f < fb ? f : fb;
      }
    }
    for (int o = 32; o >= 1; o >>= 1) {
      const double of = __shfl_xor(fb, o);
      fb = of < fb ? of : fb;
    }
    const double fthr = fb * (1.0 + rppk::FILTER_EPS);
    double bd = rpp::dinf();
    int ni = 0x7fffffff;
    for (int i0 = lane; i0 < n; i0 += 4 * TPB) {
      double vx[4], vy[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * TPB;
        const int ii = i < n ? i : i0;
        vx[u] = x[ii];
        vy[u] = y[ii];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * TPB;
        const double dx = vx[u] - rx, dy = vy[u] - ry;
        if (i < n && rpp::fast_d2(dx, dy) <= fthr) {
          const double d = rpp::py_d2(dx, dy);
          if (d < bd) {
            bd = d;
            ni = i;
          }
        }
      }
    }
    wave_argmin(bd, ni);
    s_sn += n;
    RS_M(5);
    // ---------------- steer :1541 + check_collision :1543
    int coll = 0, np0 = 0;
    const int ok0 = rs_edge(da, m, sh, x[ni], y[ni], yaw[ni], rx, ry, ryaw, pool_x + pool_used, pool_y + pool_used,
                            pool_w + pool_used, da.pool_cap - pool_used, &coll, &np0);
    if (fatal(ok0)) break;
    RS_M(6);
    s_e += ok0 ? 1 : 0;   // collision-checked edges (check_collision calls on a node), as the oracle counts them
    s_pts += ok0 ? np0 : 0;
    int nnear = -1;
    int truthy = ok0;   // `new_node` as the early-return test :1557 sees it
    if (ok0 && !coll) {
      const double nx = sh.ex, ny = sh.ey, nyaw = sh.eyaw;
      __syncthreads();
      // ---------------- find_near_nodes :1839-1863
      const double r2 = c.r2tab[n + 1];
      // stage 1: nodes whose fast distance is not clearly outside the ball, in index order (a superset of the hits)
      int kc = 0;
      const double r2hi = r2 * (1.0 + rppk::FILTER_EPS);
      for (int base0 = 0; base0 < n; base0 += 4 * TPB) {
        double vx[4], vy[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base0 + u * TPB + lane;
          const int ii = i < n ? i : 0;
          vx[u] = x[ii];
          vy[u] = y[ii];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = base0 + u * TPB + lane;
          const bool maybe = i < n && rpp::fast_d2(vx[u] - nx, vy[u] - ny) <= r2hi;
          const unsigned long long b = __ballot(maybe);
          if (maybe) mark[kc + __popcll(b & ((1ULL << lane) - 1ULL))] = i;
          kc += __popcll(b);
        }
      }
      __syncthreads();
      // stage 2: the reference's own distance for those, the <= r**2 test on it, ordered compaction
      int k = 0;
      for (int base = 0; base < kc; base += TPB) {
        const int q = base + lane;
        double d = 0.0;
        int i = 0;
        bool hit = false;
        if (q < kc) {
          i = mark[q];
          d = rpp::py_d2(x[i] - nx, y[i] - ny);
          hit = d <= r2;
        }
        const unsigned long long b = __ballot(hit);
        if (hit) {
          const int p = k + __popcll(b & ((1ULL << lane) - 1ULL));
          near[p] = i;
          ndist[p] = d;
        }
        k += __popcll(b);
      }
      s_sn += n;
      s_nh += k;
      nnear = k;
      __syncthreads();
      RS_M(7);
      // `dist_list.index(d)` :1861: an entry names the FIRST node at that distance (it is itself in the list)
      for (int base = 0; base < k; base += TPB) {
        const int p = base + lane;
        int v = -1;
        if (p < k) {
          const double d = ndist[p];
          int q = 0;
          while (q < p && ndist[q] != d) q++;
          v = near[q];
        }
        __syncthreads();
        if (p < k) near[p] = v;   // near[q], q <= p: entries before p are already final or equal to their raw value
        __syncthreads();
      }
      RS_M(8);
      // ---------------- choose_parent :1772-1813
      // The reference steers from EVERY near node and keeps the cheapest collision-free one (first in list order
      // among equal costs).  The cost of a candidate (:1901: parent cost + Euclidean distance) does not depend on
      // its steer, so the candidates are tried in ascending (cost, list position) order and the first one whose
      // edge exists and is collision free IS that minimum; the others are never steered (da.eager = 1 steers all
      // of them like the reference: same result, and RRTX_ST_REF_RAISES then also covers the edges skipped here).
      double min_cost = rpp::dinf();
      int min_ind = -1, npt = 0;
      s_ref += k;
      if (da.eager) {
        for (int p = 0; p < k; p++) {
          const int i = near[p];
          int cl = 0;
          const int tk = rs_edge(da, m, sh, x[i], y[i], yaw[i], nx, ny, nyaw, pool_x + pool_used, pool_y + pool_used,
                                 pool_w + pool_used, da.pool_cap - pool_used, &cl, &npt);
          if (fatal(tk)) break;
          s_e += tk ? 1 : 0;
          s_pts += tk ? npt : 0;
          if (tk && !cl) {
            const double cc = cost[i] + rpp::py_hypot(nx - x[i], ny - y[i]);   // Euclidean :1901-1903
            if (cc < min_cost) {
              min_cost = cc;
              min_ind = i;
            }
          }
        }
      } else {
        for (int p = lane; p < k; p += TPB) {
          const int i = near[p];
          ndist[p] = cost[i] + rpp::py_hypot(nx - x[i], ny - y[i]);
        }
        __syncthreads();
        for (;;) {
          double bv = rpp::dinf();
          int bp = 0x7fffffff;
          for (int p = lane; p < k; p += TPB) {
            const double v = ndist[p];
            if (v < bv) {
              bv = v;
              bp = p;
            }
          }
          wave_argmin(bv, bp);
          if (bp == 0x7fffffff) break;   // every candidate tried: "There is no good path" :1805
          const int i = near[bp];
          int cl = 0;
          const int tk = rs_edge(da, m, sh, x[i], y[i], yaw[i], nx, ny, nyaw, pool_x + pool_used, pool_y + pool_used,
                                 pool_w + pool_used, da.pool_cap - pool_used, &cl, &npt);
          if (fatal(tk)) break;
          s_e += tk ? 1 : 0;
          s_pts += tk ? npt : 0;
          if (tk && !cl) {
            min_cost = bv;
            min_ind = i;
            break;
          }
          if (lane == 0) ndist[bp] = rpp::dinf();
          __syncthreads();
        }
      }
      if (stop) break;
      RS_M(9);
      truthy = (k > 0 && min_cost != rpp::dinf());
      if (truthy) {
        if (n + 2 > (int)c.stride) {
          stop = 1;
          break;
        }
        if (da.eager) {
          // the re-steer of :1810 (not collision-checked again); in the lazy order the winner is the edge just laid out
          int cl = 0;
          const int bk = rs_edge(da, m, sh, x[min_ind], y[min_ind], yaw[min_ind], nx, ny, nyaw, pool_x + pool_used,
                                 pool_y + pool_used, pool_w + pool_used, da.pool_cap - pool_used, &cl, &npt);
          if (fatal(bk)) break;
          s_pts += bk ? npt : 0;
        }
        const int me = n;
        append_node(min_ind, min_cost, npt);                                  // :1811, :1548
        // ---------------- rewire :1865-1899
        // (lazy order: an edge new -> near is only steered when `improved_cost` :1888 holds at the moment of the
        // visit; a missing or colliding edge changes nothing, :1877-1878 / :1890)
        s_ref += k;
        for (int p = 0; p < k; p++) {
          if (!da.eager) {
            // next list position whose node would get cheaper; costs are re-read after every rewire
            int q = p;
            for (;;) {
              const int pp = q + lane;
              bool f = false;
              if (pp < k) {
                const int i2 = near[pp];
                f = cost[i2] > cost[me] + rpp::py_hypot(x[i2] - x[me], y[i2] - y[me]);
              }
              const unsigned long long fm = __ballot(f);
              if (fm) {
                q += __ffsll((long long)fm) - 1;
                break;
              }
              q += TPB;
              if (q >= k) break;
            }
            p = q;
            if (p >= k) break;
          }
          const int i = near[p];
          int rc2 = 0, rn = 0;
          const int ek = rs_edge(da, m, sh, x[me], y[me], yaw[me], x[i], y[i], yaw[i], pool_x + pool_used,
                                 pool_y + pool_used, pool_w + pool_used, da.pool_cap - pool_used, &rc2, &rn);
          if (fatal(ek)) break;
          s_e += ek ? 1 : 0;
          s_pts += ek ? rn : 0;
          if (!ek) continue;
          const double ec = cost[me] + rpp::py_hypot(x[i] - x[me], y[i] - y[me]);
          if (!rc2 && cost[i] > ec) {
            __syncthreads();
            if (lane == 0) {
              x[i] = sh.ex;
              y[i] = sh.ey;
              yaw[i] = sh.eyaw;
              cost[i] = ec;
              parent[i] = me;
              poff[i] = pool_used;
              plen[i] = rn;
            }
            pool_used += rn;
            s_rw++;
            // propagate_cost_to_leaves :1905-1911, level by level below node i
            RS_M(10);
            for (int j = lane; j < n; j += TPB) mark[j] = 0;
            __syncthreads();
            if (lane == 0) mark[i] = 1;
            __syncthreads();
            for (int lvl = 1;; lvl++) {
              int any = 0;
              for (int j = lane; j < n; j += TPB) {
                const int pj = parent[j];
                if (pj >= 0 && mark[pj] == lvl) {
                  cost[j] = cost[pj] + rpp::py_hypot(x[j] - x[pj], y[j] - y[pj]);
                  mark[j] = lvl + 1;
                  any++;
                }
              }
              const int tot = __popcll(__ballot(any != 0));
              for (int o = 32; o >= 1; o >>= 1) any += __shfl_xor(any, o);
              s_pr += any;
              __syncthreads();
              if (!tot) break;
              if (lvl > n) {   // cannot happen on a tree (the reference would recurse for ever); never spin on the device
                stop = 1;
                break;
              }
            }
            RS_M(11);
          }
        }
        if (stop) break;
        RS_M(10);
        // ---------------- try_goal_path :1572-1582
        int gc = 0, gn = 0;
        const int gk = rs_edge(da, m, sh, x[me], y[me], yaw[me], gx, gy, gyaw, pool_x + pool_used,
                               pool_y + pool_used, pool_w + pool_used, da.pool_cap - pool_used, &gc, &gn);
        if (fatal(gk)) break;
        s_e += gk ? 1 : 0;
        s_pts += gk ? gn : 0;
        if (gk && !gc) append_node(me, cost[me] + sh.cost_len, gn);
        RS_M(12);
      }
    }
    if (inst == c.trace_inst && lane == 0) {
      c.tr_rx[it] = rx;
      c.tr_ry[it] = ry;
      c.tr_near[it] = ni;
      c.tr_nn[it] = nnear;
    }
    __syncthreads();
    if (!c.until_max && truthy) {   // `(not search_until_max_iter) and new_node` :1557-1560
      double gb;
      int gi;
      goal_search(gb, gi);
      if (gb < rpp::dinf() && gi > 0 && gi != 0x7fffffff) done_early = 1;
    }
    if (done_early) {
      it++;   // this iteration ran
      break;
    }
  }

  // ---------------- after the loop (or the early return): search_best_goal_node, `if last_index:` :1564-1566
  if (!stop && (it >= c.max_iter || done_early)) {
    double gb;
    int gi;
    goal_search(gb, gi);
    if (lane == 0) {
      if (gb < rpp::dinf() && gi > 0 && gi != 0x7fffffff) {
        I->goal_node = gi;
        I->status |= 2;
        c.results[inst].path_cost = gb;
      }
      I->status |= 1;
    }
  }
  __syncthreads();
  for (int i = lane; i < 624; i += TPB) I->rng.mt[i] = sh.rng.mt[i];
  if (lane == 0) {
    I->rng.pos = sh.rng.pos;
    I->n = n;
    I->it = it;
    da.pool_used[inst] = pool_used;
    if (stop && !raised) I->status |= 4 | 1;              // a fixed capacity was exceeded
    if (raised) {
      I->status |= RS_ST_RAISES | 1;
      I->goal_node = raised;                               // -3 / -4: which exception
    }
    I->iterations += s_iter;
    I->edges_unique += s_e;
    I->edges_ref += da.eager ? s_e : s_ref + 2 * s_iter;   // lazy order: candidates the reference would have steered (upper bound)
    I->near_hits += s_nh;
    I->near_unique += s_nh;
    I->rewires += s_rw;
    I->propagated += s_pr;
    I->scan_nodes += s_sn;
    // algorithmic bytes: 16 B per node and scan (x, y), 24 B per polyline point produced (x, y, yaw written once)
    I->alg_bytes += 16 * s_sn + 24 * s_pts;
    I->alg_bytes2 += 16 * s_sn + 24 * s_pts;
    c.results[inst].n_nodes = n;
    c.results[inst].status = I->status;
#ifdef RRTX_PHASE_TIMERS
    for (int k = 0; k < 15; k++) I->phase[k] += sh.ph[k];   // 0 variants, 1 select, 2 course, 3 points; 4.. main-loop stages
    I->phase[15] += (int64_t)__builtin_amdgcn_s_memtime() - tk0_;   // whole kernel
#endif
  }
}

}  // namespace rppr
