// rpp_bitstar.h -- BIT* ("Batch Informed RRT*") planning core, host + device source.
// Reference: /root/reference/src_path_planning/10_path_planning_01_rrt_08_batch_informed_rrt_star.py (rrt_08)
//   RTree id codec :79-135, BITStar.setup_planning :170-207, setup_sample :209-234, plan :236-331,
//   find_final_path :333-347, remove_queue :349-357, connect :359-374, _collision_check :376-383,
//   compute_heuristic_cost / compute_distance_cost :385-395, informed_sample :397-420, sample_unit_ball :422-431,
//   best_vertex_queue_value .. best_in_edge_queue :439-474, expand_vertex :476-501, update_graph :524-556.
//   (add_vertex_to_edge_queue :503-522 never appends: its candidate is (vid, vid) and g + 0 < g is false.)
//
// BIT* here is a small, queue-driven, strictly sequential search per planning instance (<= a few hundred samples,
// <= max_iter tree vertices); the reference's results depend on Python container semantics, which are kept:
// insertion-ordered dicts (samples, tree.vertices) as ordered arrays with order-preserving deletion, queue lists with
// first-equal removal, stable sorts as first-minimum scans, the max() in best_edge_queue_value, and remove_queue's
// iteration over the list it mutates.  Parallelism is across instances (one GPU lane per instance in this first
// version); all state lives in per-instance arrays handed in through BitState.
// numpy forms as measured on the golden box: norm(v, 2) = sqrt(fma(v1, v1, v0*v0)); linspace(a, b, n)[i] =
// i * ((b - a) / (n - 1)) + a, last element = b; np.around = round-half-even; rotation C is a host input.
#pragma once
#include "rpp_core.h"

namespace rpp {

struct BitCfg {
  double start[2], goal[2];
  double rand_min, rand_max;
  double rot[4];       // C[0][0], C[0][1], C[1][0], C[1][1]
  double c_min;        // hypot(start - goal) / 1.5  (:189-190)
  double c_min2;       // c_min ** 2 (host libm)
  double num_cells;    // ceil((rand_max - rand_min) / 0.01)
  int32_t max_iter, m;
  const double *ox, *oy, *othr;
};

struct BitState {
  double *sid, *sx, *sy;            // samples (ordered dict id -> coordinates)
  double *lid, *lx, *ly;            // scratch dict of one informed_sample() batch
  double *vid, *vg, *vf, *vpar;     // tree.vertices (insertion order), g/f scores, parent id (`nodes`)
  int32_t* vhasp;
  int32_t *te_a, *te_b;             // tree edges (vertex indices) in add_edge order = adjacency list order
  double* vq;                       // vertex_queue
  double *eq_a, *eq_b;              // edge_queue
  // cached with each queue entry (values the reference recomputes on every queue scan; they never change):
  // tree index of the edge's first vertex (always a tree vertex: it comes from the vertex queue, and vertices are
  // never removed), dist(a, b), dist(b, goal); per vertex: dist(v, goal)
  int32_t *eq_ai, *vq_i;
  double *eq_dab, *eq_hb, *vh;
  int32_t *open, *closed;           // update_graph work lists
  double* path;                     // result, start -> goal
  double *tr_a, *tr_b;              // optional trace of popped edges
  int32_t scap, lcap, vcap, tecap, vqcap, eqcap, path_cap, tr_cap;
  int32_t ns, nv, nte, nvq, neq, path_n, tr_n;
  int32_t error;                    // 1: the reference would raise IndexError (empty queue); 2: capacity exceeded
  int32_t iterations, found_goal;
  double g_goal;
};

RPP_HD static inline double bit_id(const BitCfg& c, double x, double y) {     // real_world_to_node_id :79-113
  const double c0 = __builtin_rint((x - c.rand_min) / 0.01), c1 = __builtin_rint((y - c.rand_min) / 0.01);
  return 0 + c1 * c.num_cells + c0 * 1;
}
RPP_HD static inline void bit_coord(const BitCfg& c, double id, double* x, double* y) {   // :115-135
  const double c1 = __builtin_floor(id / c.num_cells);
  id = id - (c1 * c.num_cells);
  const double c0 = __builtin_floor(id / 1);
  *x = c.rand_min + 0.01 * c0;
  *y = c.rand_min + 0.01 * c1;
}
RPP_HD static inline double bit_norm(double d0, double d1) { return __builtin_sqrt(__builtin_fma(d1, d1, d0 * d0)); }
RPP_HD static inline double bit_dist(const BitCfg& c, double a, double b) {   // :385-395
  double ax, ay, bx, by;
  bit_coord(c, a, &ax, &ay);
  bit_coord(c, b, &bx, &by);
  return bit_norm(bx - ax, by - ay);
}
RPP_HD static inline int bit_dict_find(const double* ids, int n, double id) {
  for (int i = 0; i < n; i++)
    if (ids[i] == id) return i;
  return -1;
}
// dict[id] = (x, y): existing keys keep their position
RPP_HD static inline bool bit_dict_put(double* ids, double* xs, double* ys, int32_t* n, int cap, double id, double x,
                                       double y) {
  int i = bit_dict_find(ids, *n, id);
  if (i < 0) {
    if (*n >= cap) return false;
    i = (*n)++;
    ids[i] = id;
  }
  xs[i] = x;
  ys[i] = y;
  return true;
}
RPP_HD static inline int bit_vfind(const BitState& s, double id) { return bit_dict_find(s.vid, s.nv, id); }

// informed_sample(m, cMax, ...) :397-420 followed by self.samples.update(...)
template <class RNG>
RPP_HD static inline void bit_informed_sample(const BitCfg& c, BitState& s, RNG* rng, int mm, double c_max) {
  int32_t nl = 0;
  for (int i = 0; i < mm + 1; i++) {
    double rx, ry;
    if (c_max < dinf()) {
      const double r0 = c_max / 2.0;
      const double r1 = __builtin_sqrt(py_sq(c_max) - c.c_min2) / 2.0;
      double a = mt_random(rng), b = mt_random(rng);
      if (b < a) {
        const double t = a;
        a = b;
        b = t;
      }
      const double ang = 2 * 3.141592653589793 * a / b;
      const double s0 = b * rpp_glibc_cos(ang), s1 = b * rpp_glibc_sin(ang);
      const double t00 = c.rot[0] * r0, t01 = c.rot[1] * r1, t10 = c.rot[2] * r0, t11 = c.rot[3] * r1;
      rx = __builtin_fma(t00, s0, t01 * s1) + (c.start[0] + c.goal[0]) / 2.0;
      ry = __builtin_fma(t10, s0, t11 * s1) + (c.start[1] + c.goal[1]) / 2.0;
    } else {
      rx = mt_uniform(rng, c.rand_min, c.rand_max);   // sample_free_space :433-437
      ry = mt_uniform(rng, c.rand_min, c.rand_max);
    }
    if (!bit_dict_put(s.lid, s.lx, s.ly, &nl, s.lcap, bit_id(c, rx, ry), rx, ry)) s.error = 2;
  }
  for (int i = 0; i < nl; i++)
    if (!bit_dict_put(s.sid, s.sx, s.sy, &s.ns, s.scap, s.lid[i], s.lx[i], s.ly[i])) s.error = 2;
}

// edge_queue.remove at position ri (order preserving), cached columns included
RPP_HD static inline void bit_eq_erase(BitState& s, int ri) {
  for (int j = ri; j + 1 < s.neq; j++) {
    s.eq_a[j] = s.eq_a[j + 1];
    s.eq_b[j] = s.eq_b[j + 1];
    s.eq_ai[j] = s.eq_ai[j + 1];
    s.eq_dab[j] = s.eq_dab[j + 1];
    s.eq_hb[j] = s.eq_hb[j + 1];
  }
  s.neq--;
}

RPP_HD static inline double bit_g(const BitState& s, const BitCfg& c, double id, double goal_id) {
  const int vi = bit_vfind(s, id);
  if (vi >= 0) return s.vg[vi];
  return id == goal_id ? s.g_goal : dinf();
}

// BITStar(...).plan(animation=False)
template <class RNG>
RPP_HD static inline void bitstar_plan(const BitCfg& c, BitState& s, RNG* rng) {
  s.ns = s.nv = s.nte = s.nvq = s.neq = s.path_n = s.tr_n = 0;
  s.error = 0;
  s.iterations = 0;
  s.found_goal = 0;
  s.g_goal = dinf();
  // ---- setup_planning :170-207
  const double start_id = bit_id(c, c.start[0], c.start[1]), goal_id = bit_id(c, c.goal[0], c.goal[1]);
  bit_dict_put(s.sid, s.sx, s.sy, &s.ns, s.scap, goal_id, c.goal[0], c.goal[1]);
  s.vid[0] = start_id;
  s.vg[0] = 0.0;
  s.vf[0] = bit_dist(c, start_id, goal_id);
  s.vh[0] = bit_dist(c, start_id, goal_id);
  s.vhasp[0] = 0;
  s.vpar[0] = -1.0;
  s.nv = 1;
  bit_informed_sample(c, s, rng, 200, s.g_goal);
  long guard = 0;
  int seen0 = 0;
  while (s.iterations < c.max_iter && s.error == 0) {
    if (++guard > 4000000) {
      s.error = 2;
      break;
    }
    // ---- setup_sample :209-234
    if (s.nvq == 0 && s.neq == 0) {
      if (s.iterations == 0) {
        // samples are added only `if iterations != 0` (:215): a second arrival here before any edge has connected finds the
        // tree, the samples and the RNG as they were the first time -- the reference repeats that round for ever (error 3)
        if (seen0) {
          s.error = 3;
          break;
        }
        seen0 = 1;
      }
      if (s.iterations != 0) {
        int mm = 100;
        if (s.found_goal) {
          mm = 200;
          s.ns = 0;
          bit_dict_put(s.sid, s.sx, s.sy, &s.ns, s.scap, goal_id, c.goal[0], c.goal[1]);
        }
        bit_informed_sample(c, s, rng, mm, s.g_goal);
      }
      for (int q = 0; q < s.nv; q++) {
        if (bit_dict_find(s.vq, s.nvq, s.vid[q]) < 0) {
          if (s.nvq >= s.vqcap) {
            s.error = 2;
            break;
          }
          s.vq_i[s.nvq] = q;
          s.vq[s.nvq++] = s.vid[q];
        }
      }
    }
    // ---- while best_vertex_queue_value() <= best_edge_queue_value(): expand_vertex(best_in_vertex_queue()) :249-251
    for (;;) {
      double bv = dinf();
      int bvi = -1;
      for (int j = 0; j < s.nvq; j++) {
        const double val = s.vg[s.vq_i[j]] + s.vh[s.vq_i[j]];   // g(v) + h(v) :439-446
        if (val < bv) {
          bv = val;
          bvi = j;
        }
      }
      double be = dinf();
      if (s.neq) {
        be = -dinf();
        for (int j = 0; j < s.neq; j++) {   // values.sort(reverse=True)[0]: the MAXIMUM (:452-453)
          const double val = s.vg[s.eq_ai[j]] + s.eq_dab[j] + s.eq_hb[j];   // g(a) + c(a,b) + h(b) :448-456
          if (val > be) be = val;
        }
      }
      if (!(bv <= be)) break;
      if (s.nvq == 0) {
        s.error = 1;   // IndexError in best_in_vertex_queue
        return;
      }
      if (bvi < 0) bvi = 0;
      const double vid = s.vq[bvi];
      const int vidx = s.vq_i[bvi];
      for (int j = bvi; j + 1 < s.nvq; j++) {   // vertex_queue.remove(vid)
        s.vq[j] = s.vq[j + 1];
        s.vq_i[j] = s.vq_i[j + 1];
      }
      s.nvq--;
      const double d_sv = bit_dist(c, start_id, vid);
      double cx, cy;
      bit_coord(c, vid, &cx, &cy);
      for (int k = 0; k < s.ns; k++) {   // samples.items() in dict order; RAW sample coordinates (:485-488)
        if (bit_norm(s.sx[k] - cx, s.sy[k] - cy) <= 2.0 && s.sid[k] != vid) {
          const double sid = s.sid[k];
          const double d_sg = bit_dist(c, sid, goal_id), d_vs = bit_dist(c, vid, sid);
          const double est = d_sv + d_sg + d_vs;
          if (est < s.g_goal) {
            if (s.neq >= s.eqcap) {
              s.error = 2;
              return;
            }
            s.eq_a[s.neq] = vid;
            s.eq_b[s.neq] = sid;
            s.eq_ai[s.neq] = vidx;
            s.eq_dab[s.neq] = d_vs;
            s.eq_hb[s.neq] = d_sg;
            s.neq++;
          }
        }
      }
    }
    // ---- bestEdge = best_in_edge_queue(); edge_queue.remove(bestEdge) :253-255
    if (s.neq == 0) {
      s.error = 1;
      return;
    }
    int bi = 0;
    double bval = dinf();
    for (int j = 0; j < s.neq; j++) {
      const double val = s.vg[s.eq_ai[j]] + s.eq_dab[j] + s.eq_hb[j];
      if (val < bval) {
        bval = val;
        bi = j;
      }
    }
    const double ea = s.eq_a[bi], eb = s.eq_b[bi];
    const int ea_i = s.eq_ai[bi];
    const double ea_dab = s.eq_dab[bi], ea_hb = s.eq_hb[bi];
    if (s.tr_a && s.tr_n < s.tr_cap) {
      s.tr_a[s.tr_n] = ea;
      s.tr_b[s.tr_n] = eb;
    }
    s.tr_n++;
    {
      int ri = 0;
      for (int j = 0; j < s.neq; j++)
        if (s.eq_a[j] == ea && s.eq_b[j] == eb) {
          ri = j;
          break;
        }
      bit_eq_erase(s, ri);
    }
    const int v0 = ea_i;
    const double dab = ea_dab, hb = ea_hb;
    const double est_v = s.vg[v0] + dab + hb;
    const double est_e = bit_dist(c, start_id, ea) + dab + hb;
    const double act_e = s.vg[v0] + dab;
    if (est_v < s.g_goal && est_e < s.g_goal && act_e < s.g_goal) {   // f1 and f2 and f3 :270-273
      double fx, fy, tx, ty;
      bit_coord(c, ea, &fx, &fy);
      bit_coord(c, eb, &tx, &ty);
      // connect :359-374
      const int steps = (int)(bit_dist(c, bit_id(c, fx, fy), bit_id(c, tx, ty)) * 10);
      const double last_edge = bit_id(c, tx, ty);
      int npth = 0;
      double lx = 0.0, ly = 0.0;
      if (steps > 0) {
        const double stepx = steps > 1 ? (tx - fx) / (steps - 1) : 0.0, stepy = steps > 1 ? (ty - fy) / (steps - 1) : 0.0;
        npth = steps;
        for (int i = 0; i < steps; i++) {
          double px, py;
          if (steps > 1 && i == steps - 1) {
            px = tx;
            py = ty;
          } else {
            px = (stepx == 0.0 && steps > 1) ? ((double)i / (steps - 1)) * (tx - fx) + fx : i * stepx + fx;
            py = (stepy == 0.0 && steps > 1) ? ((double)i / (steps - 1)) * (ty - fy) + fy : i * stepy + fy;
          }
          bool col = false;
          for (int k = 0; k < c.m; k++) {
            const double dx = c.ox[k] - px, dy = c.oy[k] - py;
            if (dx * dx + dy * dy <= c.othr[k]) {
              col = true;
              break;
            }
          }
          if (col) {
            npth = i;
            break;
          }
          lx = px;
          ly = py;
        }
      }
      if (npth == 0) continue;   // path None or empty: no iteration count (:283-284)
      const double next_id = bit_id(c, lx, ly);
      if (bit_vfind(s, next_id) >= 0) continue;   // :291-292
      {   // del self.samples[next_id]
        const int di = bit_dict_find(s.sid, s.ns, next_id);
        if (di >= 0) {
          for (int j = di; j + 1 < s.ns; j++) {
            s.sid[j] = s.sid[j + 1];
            s.sx[j] = s.sx[j + 1];
            s.sy[j] = s.sy[j + 1];
          }
          s.ns--;
        }
      }
      if (s.nv >= s.vcap || s.nvq >= s.vqcap || s.nte >= s.tecap) {
        s.error = 2;
        return;
      }
      const int vn = s.nv++;
      s.vid[vn] = next_id;
      s.vhasp[vn] = 0;
      s.vpar[vn] = -1.0;
      s.vh[vn] = bit_dist(c, next_id, goal_id);
      s.vq_i[s.nvq] = vn;
      s.vq[s.nvq++] = next_id;
      if (next_id == goal_id || ea == goal_id) s.found_goal = 1;   // :300-303 (bestEdge rebound to (e0, next) :289)
      {   // tree.add_edge :62-66
        bool dup = false;
        for (int j = 0; j < s.nte; j++)
          if (s.te_a[j] == v0 && s.te_b[j] == vn) dup = true;
        (void)dup;   // a vertex is new here, so the pair cannot exist; adjacency always grows
        s.te_a[s.nte] = v0;
        s.te_b[s.nte] = vn;
        s.nte++;
      }
      const double gs = bit_dist(c, ea, next_id);
      s.vg[vn] = gs + s.vg[v0];
      s.vf[vn] = gs + bit_dist(c, next_id, goal_id);
      if (next_id == goal_id) s.g_goal = s.vg[vn];
      // ---- update_graph :524-556
      {
        int no = 0, ncl = 0;
        s.open[no++] = 0;
        while (no) {
          int bj = 0;
          for (int j = 1; j < no; j++)
            if (s.vf[s.open[j]] < s.vf[s.open[bj]]) bj = j;   // min(openSet, key=f): first minimum
          const int cur = s.open[bj];
          for (int j = bj; j + 1 < no; j++) s.open[j] = s.open[j + 1];
          no--;
          if (s.vid[cur] == goal_id) break;
          {
            bool in = false;
            for (int j = 0; j < ncl; j++)
              if (s.closed[j] == cur) in = true;
            if (!in) s.closed[ncl++] = cur;
          }
          for (int e = 0; e < s.nte; e++) {   // tree.vertices[cur] in append order
            int su;
            if (s.te_a[e] == cur)
              su = s.te_b[e];
            else if (s.te_b[e] == cur)
              su = s.te_a[e];
            else
              continue;
            bool in = false;
            for (int j = 0; j < ncl; j++)
              if (s.closed[j] == su) in = true;
            if (in) continue;
            const double gsc = s.vg[cur] + bit_dist(c, s.vid[cur], s.vid[su]);
            bool ino = false;
            for (int j = 0; j < no; j++)
              if (s.open[j] == su) ino = true;
            if (!ino)
              s.open[no++] = su;
            else if (gsc >= s.vg[su])
              continue;
            s.vg[su] = gsc;
            s.vf[su] = gsc + bit_dist(c, s.vid[su], goal_id);
            s.vpar[su] = s.vid[cur];
            s.vhasp[su] = 1;
            if (s.vid[su] == goal_id) s.g_goal = gsc;
          }
        }
      }
      // ---- remove_queue(lastEdge, bestEdge) :349-357 (iterates the list it mutates)
      {
        int i = 0;
        while (i < s.neq) {
          const double e1 = s.eq_b[i];
          i++;
          if (e1 == next_id) {
            if (bit_g(s, c, e1, goal_id) + 0.0 >= s.g_goal) {
              int ri = -1;
              for (int j = 0; j < s.neq; j++)
                if (s.eq_a[j] == last_edge && s.eq_b[j] == next_id) {
                  ri = j;
                  break;
                }
              if (ri >= 0) bit_eq_erase(s, ri);
            }
          }
        }
      }
    } else {   // "Nothing good" :322-325
      s.neq = 0;
      s.nvq = 0;
    }
    s.iterations++;
  }
  if (s.error) return;
  // ---- find_final_path :333-347
  int np = 0;
  bool ok = true;
  auto push = [&](double a, double b) {
    if (np < s.path_cap) {
      s.path[2 * np] = a;
      s.path[2 * np + 1] = b;
    }
    np++;
  };
  push(c.goal[0], c.goal[1]);
  double cur = goal_id;
  int hops = 0;
  while (cur != start_id) {
    double cx, cy;
    bit_coord(c, cur, &cx, &cy);
    push(cx, cy);
    const int vi = bit_vfind(s, cur);
    if (vi < 0 || !s.vhasp[vi] || ++hops > s.vcap + 2) {
      ok = false;   // KeyError: "cannot find Path" -> []
      break;
    }
    cur = s.vpar[vi];
  }
  if (ok) {
    push(c.start[0], c.start[1]);
    if (np > s.path_cap) {
      s.error = 2;
      return;
    }
    for (int i = 0; i < np / 2; i++) {   // plan[::-1]
      const int j = np - 1 - i;
      const double t0 = s.path[2 * i], t1 = s.path[2 * i + 1];
      s.path[2 * i] = s.path[2 * j];
      s.path[2 * i + 1] = s.path[2 * j + 1];
      s.path[2 * j] = t0;
      s.path[2 * j + 1] = t1;
    }
    s.path_n = np;
  } else {
    s.path_n = 0;
  }
}

}  // namespace rpp
