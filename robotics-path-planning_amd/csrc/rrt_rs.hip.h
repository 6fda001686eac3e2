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
//   * set_path's order-dependent filtering and the final arg-min are a short scan over those 48 records: lane 0
//     (rpp::rs_select), which then lays out the course (segment origins, np.arange counts: rpp::rs_course);
//   * the course's points are independent once the segment origins are known: all lanes generate them
//     (rpp::rs_point), test them against the LDS obstacle tile as they are produced (:1748-1762) and write them
//     behind the instance's polyline pool, where they stay if the edge is kept.
// choose_parent / rewire walk the near list in the reference's order, one cooperative edge after the other, so
// repeated entries of near_inds (`dist_list.index`, :1861) and nodes moved by an earlier rewire are seen exactly
// as the reference sees them.  Cost propagation is a level-synchronous sweep over the parent array.
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

struct ShR {
  rpp::MT rng;
  double ox[MAX_OBS], oy[MAX_OBS], othr[MAX_OBS];
  double vd[48][5];
  char vct[48][8];
  int32_t vst[48], vn[48];
  rpp::RsCourse course;
  double cost_len;           // sum(|course_lengths|) of the chosen path (:1600)
  double ex, ey, eyaw;       // last point of the course = pose of the node steer() returns (:1592-1594)
  double rx, ry, ryaw;
  int32_t sel, any_hit, flag;
};

// Cooperative steer (fx,fy,fyaw) -> (tx,ty,tyaw).  Wave-uniform return: 1 a node exists (course in sh.course, its
// points written at px/py/pyaw when they fit `room`, *coll = check_collision fails, *npts = points), 0 steer
// returns None, < 0 the reference raises (-3 ZeroDivisionError, -4 ValueError), -1 pool full.
__device__ __noinline__ int rs_edge(const DubArgs& da, int m, ShR& sh, double fx, double fy, double fyaw, double tx,
                                    double ty, double tyaw, double* __restrict__ px, double* __restrict__ py,
                                    double* __restrict__ pyaw, int64_t room, int* coll, int* npts) {
  const int lane = threadIdx.x;
  if (lane < 48) {
    rpp::RsFrame F;
    rpp::rs_frame(fx, fy, fyaw, tx, ty, tyaw, da.curvature, da.step_size, &F);
    int nn = 0;
    sh.vst[lane] = rpp::rs_variant(lane >> 2, lane & 3, F, sh.vd[lane], sh.vct[lane], &nn);
    sh.vn[lane] = nn;
  }
  __syncthreads();
  if (lane == 0) {
    const int sel = rpp::rs_select(sh.vst, sh.vd, sh.vct, sh.vn, da.step_size * da.curvature, da.curvature);
    sh.sel = sel;
    sh.any_hit = 0;
    if (sel >= 0) {
      rpp::rs_course(sh.vd[sel], sh.vct[sel], sh.vn[sel], fx, fy, fyaw, da.curvature, da.step_size, &sh.course);
      double s = 0.0;
      for (int i = 0; i < sh.vn[sel]; i++) s += rpp::dabs(sh.vd[sel][i] / da.curvature);
      sh.cost_len = s;
    }
  }
  __syncthreads();
  const int sel = sh.sel;
  if (sel < -1) return sel;
  if (sel < 0) return 0;
  const int total = sh.course.total;
  *npts = total;
  if (total <= 0) return 0;   // `if not px` :1588
  if (total > room) return -1;
  int hit = 0;
  for (int k = lane; k < total; k += TPB) {
    double wx, wy, wyaw;
    rpp::rs_point(sh.course, k, &wx, &wy, &wyaw);
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
  *coll = sh.any_hit;
  return 1;
}

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

__global__ __launch_bounds__(TPB) void rrt_rs_kernel(Ctx c, DubArgs da, int iters) {
  __shared__ ShR sh;
  const int inst = blockIdx.x;
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
  double* pool_x = da.pool_x + (int64_t)inst * da.pool_cap;
  double* pool_y = da.pool_y + (int64_t)inst * da.pool_cap;
  double* pool_w = da.pool_yaw + (int64_t)inst * da.pool_cap;
  const int m = c.m;

  for (int i = lane; i < 624; i += TPB) sh.rng.mt[i] = I->rng.mt[i];
  for (int i = lane; i < m; i += TPB) {
    sh.ox[i] = c.ox[i];
    sh.oy[i] = c.oy[i];
    sh.othr[i] = c.othr[i];
  }
  if (lane == 0) sh.rng.pos = I->rng.pos;
  __syncthreads();
  int n = I->n, it = I->it;
  int64_t pool_used = da.pool_used[inst];
  const double gx = I->goal[0], gy = I->goal[1];
  if (it == 0 && lane == 0) {
    yaw[0] = da.start_yaw;
    poff[0] = 0;
    plen[0] = 0;
  }
  __syncthreads();
  int64_t s_iter = 0, s_e = 0, s_nh = 0, s_rw = 0, s_pr = 0, s_sn = 0;
  int stop = 0, done_early = 0, raised = 0;

  // search_best_goal_node :1815-1836: lowest cost inside both thresholds, first index among equals
  auto goal_search = [&](double& gb, int& gi) {
    double best = rpp::dinf();
    int bidx = 0x7fffffff;
    for (int i = lane; i < n; i += TPB) {
      if (rpp::py_hypot(x[i] - gx, y[i] - gy) <= da.goal_xy_th && rpp::dabs(yaw[i] - da.goal_yaw) <= da.goal_yaw_th) {
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

  for (int step = 0; step < iters && it < c.max_iter && !stop; step++, it++) {
    s_iter++;
    // ---------------- get_random_node :1658-1666
    if (lane == 0) {
      sh.rx = rpp::mt_uniform(&sh.rng, c.rand_min, c.rand_max);
      sh.ry = rpp::mt_uniform(&sh.rng, c.rand_min, c.rand_max);
      sh.ryaw = rpp::mt_uniform(&sh.rng, -rpp::kPi, rpp::kPi);
    }
    __syncthreads();
    const double rx = sh.rx, ry = sh.ry, ryaw = sh.ryaw;
    // ---------------- get_nearest_node_index :1728-1734 (x, y only; first minimum)
    double bd = rpp::dinf();
    int ni = 0x7fffffff;
    for (int i = lane; i < n; i += TPB) {
      const double d = rpp::py_d2(x[i] - rx, y[i] - ry);
      if (d < bd) {
        bd = d;
        ni = i;
      }
    }
    wave_argmin(bd, ni);
    s_sn += n;
    // ---------------- steer :1541 + check_collision :1543
    int coll = 0, np0 = 0;
    const int ok0 = rs_edge(da, m, sh, x[ni], y[ni], yaw[ni], rx, ry, ryaw, pool_x + pool_used, pool_y + pool_used,
                            pool_w + pool_used, da.pool_cap - pool_used, &coll, &np0);
    if (fatal(ok0)) break;
    s_e++;
    int nnear = -1;
    int truthy = ok0;   // `new_node` as the early-return test :1557 sees it
    if (ok0 && !coll) {
      const double nx = sh.ex, ny = sh.ey, nyaw = sh.eyaw;
      __syncthreads();
      // ---------------- find_near_nodes :1839-1863
      const double r2 = c.r2tab[n + 1];
      int k = 0;
      for (int base = 0; base < n; base += TPB) {
        const int i = base + lane;
        double d = 0.0;
        bool hit = false;
        if (i < n) {
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
      // ---------------- choose_parent :1772-1813
      double min_cost = rpp::dinf();
      int min_ind = -1;
      for (int p = 0; p < k; p++) {
        const int i = near[p];
        int cl = 0, npt = 0;
        const int tk = rs_edge(da, m, sh, x[i], y[i], yaw[i], nx, ny, nyaw, pool_x + pool_used, pool_y + pool_used,
                               pool_w + pool_used, da.pool_cap - pool_used, &cl, &npt);
        if (fatal(tk)) break;
        s_e++;
        if (tk && !cl) {
          const double cc = cost[i] + rpp::py_hypot(nx - x[i], ny - y[i]);   // Euclidean :1901-1903
          if (cc < min_cost) {
            min_cost = cc;
            min_ind = i;
          }
        }
      }
      if (stop) break;
      truthy = (k > 0 && min_cost != rpp::dinf());
      if (truthy) {
        if (n + 2 > (int)c.stride) {
          stop = 1;
          break;
        }
        int cl = 0, npt = 0;
        const int bk = rs_edge(da, m, sh, x[min_ind], y[min_ind], yaw[min_ind], nx, ny, nyaw, pool_x + pool_used,
                               pool_y + pool_used, pool_w + pool_used, da.pool_cap - pool_used, &cl, &npt);   // :1810
        if (fatal(bk)) break;
        s_e++;
        const int me = n;
        append_node(min_ind, min_cost, npt);                                  // :1811, :1548
        // ---------------- rewire :1865-1899
        for (int p = 0; p < k; p++) {
          const int i = near[p];
          int rc2 = 0, rn = 0;
          const int ek = rs_edge(da, m, sh, x[me], y[me], yaw[me], x[i], y[i], yaw[i], pool_x + pool_used,
                                 pool_y + pool_used, pool_w + pool_used, da.pool_cap - pool_used, &rc2, &rn);
          if (fatal(ek)) break;
          s_e++;
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
          }
        }
        if (stop) break;
        // ---------------- try_goal_path :1572-1582
        int gc = 0, gn = 0;
        const int gk = rs_edge(da, m, sh, x[me], y[me], yaw[me], gx, gy, da.goal_yaw, pool_x + pool_used,
                               pool_y + pool_used, pool_w + pool_used, da.pool_cap - pool_used, &gc, &gn);
        if (fatal(gk)) break;
        s_e++;
        if (gk && !gc) append_node(me, cost[me] + sh.cost_len, gn);
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
    I->edges_ref += s_e;
    I->near_hits += s_nh;
    I->near_unique += s_nh;
    I->rewires += s_rw;
    I->propagated += s_pr;
    I->scan_nodes += s_sn;
    c.results[inst].n_nodes = n;
    c.results[inst].status = I->status;
  }
}

}  // namespace rppr
