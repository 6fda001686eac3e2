// rrt_informed.hip.h -- Informed RRT* iteration kernel (gfx950).
// Reference: /root/reference/src_path_planning/10_path_planning_01_rrt_07_informed_rrt_star.py  (rrt_07)
//   informed_rrt_star_search :1044-1108, informed_sample :1145-1159, sample_unit_ball :1162-1171,
//   sample_free_space(_sobol) :1173-1191, get_nearest_list_index :1210-1214, get_new_node :1216-1224,
//   check_collision :1271-1276 -> check_segment_collision :1263-1269 -> distance_squared_point_to_segment :1249-1261,
//   find_near_nodes :1137-1143, choose_parent :1110-1135, rewire :1232-1246, goal bookkeeping :1094-1103.
//
// Same structure and HBM layout as rrt_kernels.hip.h (one 256-thread workgroup per instance, streaming nearest and
// near-ball passes over SoA x[], y[] with the exact-`**2` re-check, LDS obstacle tile).  Differences that matter:
//  * fixed-length step, analytic point-to-segment collision (lanes over candidate x obstacle pairs);
//  * near radius 50*sqrt(ln n / n), NOT capped: the candidate list reaches several hundred entries early on;
//  * integer parents, no cost propagation; rewire(new -> node) re-derives exactly the (theta, d) choose_parent
//    computed for (node -> new) -- hypot of the negated differences, atan2 of the same arguments -- so ONE
//    collision test per candidate serves both (the reference evaluates it twice);
//  * ellipsoidal informed sampling once a path exists: numpy's products are restated in the fused forms measured
//    on the golden-generating box: 2-vector dot = fma(u1, w1, u0*w0); (3x3)@(3x1) row =
//    fma(T_i0, x0, T_i1*x1); the rotation C (numpy SVD, rrt_07:1061-1068) is an input computed on the host.
#pragma once
#include "rrt_kernels.hip.h"

namespace rppi {

using rppk::Ctx;
using rppk::Inst;
using rppk::FILTER_EPS;

constexpr int TPB = 256;
constexpr int NW = TPB / 64;
constexpr int MAX_OBS = rppk::MAX_OBS;
// NUI distinct near candidates held in LDS: 512 in the product shape (38 KB, 4 workgroups per CU); instances whose near
// set outgrows it (the radius of rrt_07:1139 is not capped, so a tree confined to a pocket sees most of itself) are
// planned again on the 2048-candidate shape (one workgroup per CU) -- rrtx_api.hip.
constexpr int NUI_SMALL = 512, NUI_LARGE = 2048;

template <int NUI>
struct ShIT {
  static constexpr int kTPB = TPB, kNW = NW;   // shape of the rppk:: helpers
  rpp::MT rng;
  double ox[MAX_OBS], oy[MAX_OBS], othr[MAX_OBS];
  double orad[MAX_OBS];   // obstacle radius (sqrt of the threshold), rounded up: the cheap reject test of choose_parent
  int32_t uidx[NUI], ufree[NUI];
  // uval (d**2 of a candidate, duplicate collapse) and cval (per-thread scratch of the same phase) are dead before
  // choose_parent writes ud / uex: they share storage, which keeps the block under 40 KB (4 workgroups per CU)
  union { double uval[NUI]; double ud[NUI]; };
  union { double cval[TPB]; double uex[NUI]; };
  double ux[NUI], uy[NUI], ucost[NUI], uey[NUI];
  int32_t cflag[TPB];
  double red_best[NW], red_second[NW];
  int32_t red_idx[NW], wave_cnt[NW], wave_start[NW];
  double rx, ry, nx, ny, ex, ey, ncost, cbest, plen;
  int32_t flag, nu, nvalid, overflow, ecoll, npar, nrw;
};

// numpy `u.dot(w)` for 2-vectors on the golden box: fma(u1, w1, u0*w0)
__device__ __forceinline__ double np_dot2(double u0, double u1, double w0, double w1) {
  return __builtin_fma(u1, w1, u0 * w0);
}
// distance_squared_point_to_segment (rrt_07:1249-1261)
__device__ __forceinline__ double seg_dist2(double vx, double vy, double wx, double wy, double px, double py) {
  if (vx == wx && vy == wy) {
    const double a = px - vx, b = py - vy;
    return np_dot2(a, b, a, b);
  }
  const double wv0 = wx - vx, wv1 = wy - vy;
  const double l2 = np_dot2(wv0, wv1, wv0, wv1);
  const double pv0 = px - vx, pv1 = py - vy;
  const double tt = np_dot2(pv0, pv1, wv0, wv1) / l2;
  double t = tt < 1.0 ? tt : 1.0;   // min(1, tt)
  t = t > 0.0 ? t : 0.0;            // max(0, .)
  const double pr0 = vx + t * wv0, pr1 = vy + t * wv1;
  const double q0 = px - pr0, q1 = py - pr1;
  return np_dot2(q0, q1, q0, q1);
}

// exact re-check + `.index` de-dup (rrt_07:1140-1142) into the candidate records (idx, d2, x, y, cost)
template <int NUI>
__device__ __forceinline__ void build_candidates_i(const double* __restrict__ x, const double* __restrict__ y,
                                                   const double* __restrict__ cost, double qx, double qy,
                                                   double thr_exact, const int32_t* hits, int kraw, ShIT<NUI>& sh) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) {
    sh.nu = 0;
    sh.nvalid = 0;
  }
  __syncthreads();
  for (int base = 0; base < kraw; base += TPB) {
    const int h = base + tid;
    int idx = -1;
    double v = 0.0, hx = 0.0, hy = 0.0;
    bool valid = false;
    if (h < kraw) {
      idx = rppk::hit_at(hits, sh, h);
      hx = x[idx];
      hy = y[idx];
      v = rpp::py_d2(hx - qx, hy - qy);
      valid = v <= thr_exact;
    }
    const int nu = sh.nu;
    // "an equal value is held by an earlier entry" (rppk::exact_dedup): uniform trip counts, broadcast LDS reads, no
    // early exit -- rrt_07's near sets hold hundreds of hits and these loops were the longest part of the phase
    bool dupe = false;
#pragma unroll 4
    for (int u = 0; u < nu; u++) dupe |= (sh.uval[u] == v);
    const bool cand = valid && !dupe;
    sh.cval[tid] = cand ? v : rpp::b2d(0x7ff8000000000000ULL);   // NaN: never equal
    sh.cflag[tid] = cand ? 1 : 0;
    __syncthreads();
    const int nchunk = (kraw - base) < TPB ? (kraw - base) : TPB;
    bool earlier = false;
#pragma unroll 4
    for (int t = 0; t < nchunk; t++) earlier |= (t < tid) & (sh.cval[t] == v);
    const bool first = cand && !earlier;
    const uint64_t mf = __ballot(first), mv = __ballot(valid);
    if (lane == 0) {
      sh.red_idx[w] = __popcll(mf);
      atomicAdd(&sh.nvalid, __popcll(mv));
    }
    __syncthreads();
    int off = nu;
#pragma unroll
    for (int k = 0; k < NW; k++)
      if (k < w) off += sh.red_idx[k];
    int tot = nu;
#pragma unroll
    for (int k = 0; k < NW; k++) tot += sh.red_idx[k];
    if (first) {
      const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int p = off + __popcll(mf & lt_mask);
      if (p < NUI) {
        sh.uval[p] = v;
        sh.uidx[p] = idx;
        sh.ux[p] = hx;
        sh.uy[p] = hy;
        sh.ucost[p] = cost[idx];
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (tot > NUI) {
        sh.overflow = 1;
        tot = NUI;
      }
      sh.nu = tot;
    }
    __syncthreads();
  }
}

// check_collision (rrt_07:1271-1276) of the candidate slots listed in sh.cflag[0..nt) (nt <= TPB): lane t derives
// (theta, end point) of its candidate exactly as choose_parent :1117-1119 / rewire :1242-1244 do -- both arrive at the same
// (theta, d) -- then a wave takes a candidate, its lanes the obstacles.  The exact segment distance (:1249-1261) is
// evaluated only for obstacles that can touch the segment: dist(p, segment) >= |p - midpoint| - half length, so
// |p - mid| > half length + radius (with 1e-9 of slack against the roundings of this test) leaves
// `distance**2 <= size**2` false whatever the exact form returns.  Result in sh.ufree[e] (1 free, 0 blocked).
template <int NUI>
__device__ __forceinline__ void test_candidates(const Ctx& c, ShIT<NUI>& sh, int nt, double nx, double ny) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < nt) {
    const int e = sh.cflag[tid];
    const double dx = nx - sh.ux[e], dy = ny - sh.uy[e];
    const double th = rpp_glibc_atan2(dy, dx);
    const double d = sh.ud[e];
    sh.uex[e] = sh.ux[e] + rpp_glibc_cos(th) * d;
    sh.uey[e] = sh.uy[e] + rpp_glibc_sin(th) * d;
  }
  __syncthreads();
  for (int t = w; t < nt; t += NW) {
    const int e = sh.cflag[t];
    const double vx = sh.ux[e], vy = sh.uy[e], ex2 = sh.uex[e], ey2 = sh.uey[e];
    const double mx = 0.5 * (vx + ex2), my = 0.5 * (vy + ey2), hl = 0.5 * sh.ud[e] * (1.0 + 1e-12);
    bool hit = false;
    for (int k = lane; k < c.m; k += 64) {
      const double ddx = sh.ox[k] - mx, ddy = sh.oy[k] - my, tt = hl + sh.orad[k];
      if (ddx * ddx + ddy * ddy <= tt * tt * (1.0 + 1e-9)) {
        if (seg_dist2(vx, vy, ex2, ey2, sh.ox[k], sh.oy[k]) <= sh.othr[k]) hit = true;
      }
    }
    const bool any = __ballot(hit) != 0ull;
    if (lane == 0) sh.ufree[e] = any ? 0 : 1;
  }
  __syncthreads();
}

// per instance -- rot: C[0][0], C[0][1], C[1][0], C[1][1]; xc: ellipse centre; c_min2 = c_min**2 (host libm)
struct InformedArgs {
  double rot[4];
  double xc[2];
  double c_min2;
};

template <int NUI, int WPS>
__global__ __launch_bounds__(TPB, WPS) void rrt_informed_kernel(Ctx c, const InformedArgs* __restrict__ per_inst,
                                                                double* cbest_io, int iters, int eager) {
  __shared__ ShIT<NUI> sh;
  const int inst = c.inst_map ? c.inst_map[blockIdx.x] : blockIdx.x;
  const InformedArgs ia = per_inst[inst];   // rotation, centre and c_min**2 of THIS instance's start / goal pair
  const int tid = threadIdx.x;
  Inst* I = c.inst + inst;
  if (I->status & 1) return;
  const int64_t off = (int64_t)inst * c.stride;
  double* __restrict__ x = c.x + off;
  double* __restrict__ y = c.y + off;
  double* __restrict__ cost = c.cost + off;
  int32_t* parent = c.parent + off;
  int32_t* hits = c.hits + off;
  double* pathbuf = c.path_xy + (int64_t)inst * c.path_cap * 2;
  // f32 mirror of the coordinates for the two streaming passes (see scan_nearest_f32): valid while every coordinate
  // and query stays below the magnitude the margin c.f32_m was derived from; informed samples are not clipped to the
  // sampling square (rrt_07:1145-1159), so an instance that leaves it switches to the f64 passes for good
  float* __restrict__ xf = c.xf ? c.xf + off : nullptr;
  float* __restrict__ yf = c.xf ? c.yf + off : nullptr;
  const double fm = c.f32_m, fmax = c.f32_m * 1048576.0;   // margin = 2^-20 * fmax
  int f32_ok = (c.xf != nullptr) && (I->first_goal != -3);

  for (int i = tid; i < 624; i += TPB) sh.rng.mt[i] = I->rng.mt[i];
  for (int i = tid; i < c.m; i += TPB) {
    sh.ox[i] = c.ox[i];
    sh.oy[i] = c.oy[i];
    sh.othr[i] = c.othr[i];
    sh.orad[i] = __builtin_sqrt(c.othr[i] > 0.0 ? c.othr[i] : 0.0) * (1.0 + 1e-12);
  }
  if (tid == 0) {
    sh.rng.pos = I->rng.pos;
    sh.overflow = 0;
    sh.cbest = cbest_io[inst];
  }
  __syncthreads();
  int n = I->n, it = I->it;
  const double gx = I->goal[0], gy = I->goal[1], sx0 = I->start[0], sy0 = I->start[1];
  const double E = c.expand_dis;
  rpp::Sobol sob = I->sobol;
  int64_t s_iter = 0, s_eu = 0, s_er = 0, s_nh = 0, s_nu = 0, s_rw = 0, s_sn = 0, s_ab = 0, s_ab2 = 0, s_ex = 0;
  int stop = 0;
  PH_DECL

  for (int step = 0; step < iters && it < c.max_iter && !stop; step++, it++) {
    s_iter++;
    PH(15);
    // ---------------- informed_sample :1145-1159
    if (tid == 0) {
      double rx, ry;
      const double cb = sh.cbest;
      if (cb < rpp::dinf()) {
        const double r0 = cb / 2.0;
        const double r1 = __builtin_sqrt(rpp::py_sq(cb) - ia.c_min2) / 2.0;          // :1147
        double a = rpp::mt_random(&sh.rng), b = rpp::mt_random(&sh.rng);            // sample_unit_ball :1162-1171
        if (b < a) {
          const double t = a;
          a = b;
          b = t;
        }
        const double ang = 2 * 3.141592653589793 * a / b;
        const double s0 = b * rpp_glibc_cos(ang), s1 = b * rpp_glibc_sin(ang);
        const double t00 = ia.rot[0] * r0, t01 = ia.rot[1] * r1, t10 = ia.rot[2] * r0, t11 = ia.rot[3] * r1;
        rx = __builtin_fma(t00, s0, t01 * s1) + ia.xc[0];                            // np.dot(np.dot(c, rl), x_ball) + x_center :1151
        ry = __builtin_fma(t10, s0, t11 * s1) + ia.xc[1];
      } else if (rpp::mt_randint_0_100(&sh.rng) > c.goal_sample_rate) {              // :1173-1191
        if (c.sampler == 1) {
          double q[2];
          rpp::sobol_next(&sob, q);
          rx = c.rand_min + q[0] * (c.rand_max - c.rand_min);
          ry = c.rand_min + q[1] * (c.rand_max - c.rand_min);
        } else {
          rx = rpp::mt_uniform(&sh.rng, c.rand_min, c.rand_max);
          ry = rpp::mt_uniform(&sh.rng, c.rand_min, c.rand_max);
        }
      } else {
        rx = gx;
        ry = gy;
      }
      sh.rx = rx;
      sh.ry = ry;
    }
    __syncthreads();
    const double rx = sh.rx, ry = sh.ry;

    PH(0);
    // ---------------- nearest :1210-1214
    int ni;
    double gbest, gsecond;
    const bool q_ok = f32_ok && rpp::dabs(rx) <= fmax && rpp::dabs(ry) <= fmax;
    bool nearest_done = false;
    if (q_ok) {
      int fi;
      double fb, fs;
      rppk::scan_nearest_f32(xf, yf, n, (float)rx, (float)ry, sh, fi, fb, fs);
      s_ab += 8 * (int64_t)n;
      if (__builtin_sqrt(fs) - __builtin_sqrt(fb) > 2.0 * fm) {
        ni = fi;
        gbest = 1.0;
        gsecond = rpp::dinf();
        nearest_done = true;
      }
    }
    if (!nearest_done) {
      rppk::scan_nearest(x, y, n, rx, ry, sh, ni, gbest, gsecond);
      s_ab += 16 * (int64_t)n;
    }
    s_sn += n;
    s_ab += 24 * (int64_t)c.m;
    s_ab2 += 16 * (int64_t)n + 24 * (int64_t)c.m;
    if (gbest != 0.0 && gsecond <= gbest * (1.0 + FILTER_EPS)) {
      s_ex++;
      const int kraw = rppk::scan_hits(x, y, n, rx, ry, gbest * (1.0 + FILTER_EPS), hits, sh);
      double best = rpp::dinf(), second = rpp::dinf();
      int bidx = 0x7fffffff;
      for (int h = tid; h < kraw; h += TPB) {
        const int idx = rppk::hit_at(hits, sh, h);
        const double d = rpp::py_d2(x[idx] - rx, y[idx] - ry);
        if (d < best || (d == best && idx < bidx)) {
          best = d;
          bidx = idx;
        }
      }
      double gb2, gs2;
      rppk::block_argmin(best, bidx, second, sh, gb2, ni, gs2);
    }

    PH(1);
    // ---------------- steer :1080-1083 / get_new_node :1216-1224, extension edge :1085
    if (tid == 0) {
      const double qx = x[ni], qy = y[ni];
      const double theta = rpp_glibc_atan2(ry - qy, rx - qx);
      const double ct = rpp_glibc_cos(theta), st = rpp_glibc_sin(theta);
      const double nx = qx + E * ct, ny = qy + E * st;
      const double d = rpp::py_hypot(qx - nx, qy - ny);        // line_cost :1206
      sh.nx = nx;
      sh.ny = ny;
      sh.ux[0] = qx;                                            // segment start (candidate slot 0 is free here)
      sh.uy[0] = qy;
      sh.ex = qx + ct * d;                                      // check_collision :1273-1274
      sh.ey = qy + st * d;
      sh.ncost = cost[ni] + E;                                  // :1222
      sh.npar = ni;
      sh.ecoll = 0;
    }
    __syncthreads();
    const double nx = sh.nx, ny = sh.ny;
    s_eu++;
    s_er++;
    for (int k = tid; k < c.m; k += TPB)
      if (seg_dist2(sh.ux[0], sh.uy[0], sh.ex, sh.ey, sh.ox[k], sh.oy[k]) <= sh.othr[k]) sh.ecoll = 1;
    __syncthreads();
    const int accepted = !sh.ecoll;
    int nnear = -1;

    if (accepted) {
      PH(2);
      // ---------------- find_near_nodes :1137-1143  (n_node = len(node_list), radius not capped)
      const double r2 = c.r2tab[n];
      int kraw;
      if (f32_ok && rpp::dabs(nx) <= fmax && rpp::dabs(ny) <= fmax) {
        const double rr = __builtin_sqrt(r2) + fm;   // ball radius r + m in the f32 metric, rounded up
        kraw = rppk::scan_hits_f32(xf, yf, n, (float)nx, (float)ny, (float)(rr * rr * (1.0 + 1e-6)), hits, sh);
        s_ab += 8 * (int64_t)n + 16 * (int64_t)kraw;
      } else {
        kraw = rppk::scan_hits(x, y, n, nx, ny, r2 * (1.0 + FILTER_EPS), hits, sh);
        s_ab += 16 * (int64_t)n;
      }
      s_sn += n;
      build_candidates_i(x, y, cost, nx, ny, r2, hits, kraw, sh);
      const int nu = sh.nu, nvalid = sh.nvalid;
      nnear = nu;
      s_nh += nvalid;
      s_nu += nu;
      s_ab += 48 * (int64_t)nu + 28;
      s_ab2 += 16 * (int64_t)n + 48 * (int64_t)nu + 28;
      PH(4);
      if (!eager) {
        // ---------------- choose_parent :1110-1135, cheapest first.  A candidate's cost `cost_i + d` (:1120) does not
        // depend on its collision test, so the candidates are ranked by (cost, list position) and tested in growing
        // batches (4, 8, 16 ...): the first free one in rank order IS the reference's `min(d_list)` / `.index` pick
        // (:1125-1126); later ones are never tested.  atan2 / cos / sin (:1118, :1273-1274) are evaluated for tested
        // candidates only.  ufree: -1 not tested, 0 blocked, 1 free.  The rank lives in uey[] until the slot is tested.
        for (int e = tid; e < nu; e += TPB) {
          sh.ud[e] = rpp::py_hypot(nx - sh.ux[e], ny - sh.uy[e]);
          sh.ufree[e] = -1;
        }
        __syncthreads();
        for (int e = tid; e < nu; e += TPB) {
          const double ce = sh.ucost[e] + sh.ud[e];
          int rk = 0;
#pragma unroll 4
          for (int j = 0; j < nu; j++) {
            const double cj = sh.ucost[j] + sh.ud[j];
            rk += (cj < ce || (cj == ce && j < e)) ? 1 : 0;
          }
          sh.uey[e] = (double)rk;
        }
        __syncthreads();
        int tested = 0, found = 0;
        for (int base = 0, bsz = 4; base < nu && !found; base += bsz, bsz = bsz < TPB / 2 ? 2 * bsz : TPB) {
          const int na = (nu - base < bsz) ? nu - base : bsz;
          if (tid == 0) sh.flag = 0x7fffffff;
          for (int e = tid; e < nu; e += TPB) {
            if (sh.ufree[e] >= 0) continue;   // tested in an earlier batch (its uey[] is an end point now)
            const int rk = (int)sh.uey[e];
            if (rk >= base && rk < base + na) sh.cflag[rk - base] = e;
          }
          __syncthreads();
          test_candidates(c, sh, na, nx, ny);   // overwrites uey[] of the tested slots only: their rank is in cflag order
          tested += na;
          if (tid < na && sh.ufree[sh.cflag[tid]] == 1) atomicMin(&sh.flag, tid);
          __syncthreads();
          const int t = sh.flag;
          if (t != 0x7fffffff) {
            found = 1;
            if (tid == 0) {
              const int e = sh.cflag[t];
              sh.ncost = sh.ucost[e] + sh.ud[e];   // :1132-1133
              sh.npar = sh.uidx[e];
            }
          }
          __syncthreads();
        }
        s_eu += tested;
        s_er += nvalid;
        const double ncost = sh.ncost;
        if (rpp::dabs(nx) > fmax || rpp::dabs(ny) > fmax) f32_ok = 0;   // outside the magnitude the margin covers
        PH(6);
        // ---------------- append :1091, rewire :1232-1246: only a candidate with `near_node.cost > s_cost` (:1241) gets a
        // collision test (:1244); nothing the loop writes is read by a later candidate (no propagation in rrt_07), so
        // the candidates are independent.  Those not tested by choose_parent are tested now.
        if (tid == 0) {
          x[n] = nx;
          y[n] = ny;
          if (xf) {
            xf[n] = (float)nx;
            yf[n] = (float)ny;
          }
          cost[n] = ncost;
          parent[n] = sh.npar;
          sh.nrw = 0;
          sh.flag = 0;
        }
        __syncthreads();
        int extra = 0;
        for (int e0 = 0; e0 < nu; e0 += TPB) {
          const int e = e0 + tid;
          const bool need = e < nu && sh.ucost[e] > ncost + sh.ud[e] && sh.ufree[e] < 0;
          if (need) sh.cflag[atomicAdd(&sh.flag, 1)] = e;
          __syncthreads();
          const int nt = sh.flag;
          __syncthreads();
          if (nt > 0) {
            test_candidates(c, sh, nt, nx, ny);
            extra += nt;
            if (tid == 0) sh.flag = 0;
            __syncthreads();
          }
        }
        int my_rw = 0, my_chk = 0;
        for (int e = tid; e < nu; e += TPB) {
          const double s_cost = ncost + sh.ud[e];
          if (sh.ucost[e] > s_cost) {          // near_node.cost > s_cost :1241
            my_chk++;
            if (sh.ufree[e] == 1) {            // check_collision(near_node, theta, d) :1244
              const int u = sh.uidx[e];
              parent[u] = n;
              cost[u] = s_cost;
              my_rw++;
            }
          }
        }
        if (my_rw) atomicAdd(&sh.nrw, my_rw);
        if (my_chk) atomicAdd(&sh.nvalid, my_chk);   // nvalid reused: rewire collision tests the reference makes
        n++;
        __syncthreads();
        s_rw += sh.nrw;
        s_eu += extra;
        s_er += sh.nvalid - nvalid;
        PH(9);
      } else {
      // ---------------- choose_parent :1110-1135 : (d, theta, end point) per candidate, then candidate x obstacle
      for (int e = tid; e < nu; e += TPB) {
        const double dx = nx - sh.ux[e], dy = ny - sh.uy[e];
        const double d = rpp::py_hypot(dx, dy);
        const double th = rpp_glibc_atan2(dy, dx);
        sh.ud[e] = d;
        sh.uex[e] = sh.ux[e] + rpp_glibc_cos(th) * d;
        sh.uey[e] = sh.uy[e] + rpp_glibc_sin(th) * d;
        sh.ufree[e] = 1;
      }
      __syncthreads();
      // candidate x obstacle (check_collision :1271-1276 per candidate): a wave takes a candidate, its lanes the
      // obstacles.  The exact segment distance (:1249-1261) is evaluated only for obstacles that can touch the segment:
      // dist(p, segment) >= |p - midpoint| - half length, so |p - mid| > half length + radius (with 1e-9 of slack
      // against the roundings of this test) leaves `distance**2 <= size**2` false whatever the exact form returns.
      {
        const int lane = tid & 63, w = tid >> 6;
        for (int e = w; e < nu; e += NW) {
          const double vx = sh.ux[e], vy = sh.uy[e], ex2 = sh.uex[e], ey2 = sh.uey[e];
          const double mx = 0.5 * (vx + ex2), my = 0.5 * (vy + ey2), hl = 0.5 * sh.ud[e] * (1.0 + 1e-12);
          bool hit = false;
          for (int k = lane; k < c.m; k += 64) {
            const double ddx = sh.ox[k] - mx, ddy = sh.oy[k] - my, t = hl + sh.orad[k];
            if (ddx * ddx + ddy * ddy <= t * t * (1.0 + 1e-9)) {
              if (seg_dist2(vx, vy, ex2, ey2, sh.ox[k], sh.oy[k]) <= sh.othr[k]) hit = true;
            }
          }
          if (__ballot(hit) != 0ull && lane == 0) sh.ufree[e] = 0;
        }
      }
      __syncthreads();
      s_eu += nu;
      s_er += nvalid;
      if (nu > 0) {
        double best = rpp::dinf(), second = rpp::dinf(), mn, gs;
        int bidx = 0x7fffffff, sel;
        for (int e = tid; e < nu; e += TPB) {
          const double cc = sh.ufree[e] ? sh.ucost[e] + sh.ud[e] : rpp::dinf();   // :1120-1123
          if (cc < best) {
            best = cc;
            bidx = e;
          }
        }
        rppk::block_argmin(best, bidx, second, sh, mn, sel, gs);                  // first minimum :1125-1126
        if (tid == 0 && mn < rpp::dinf()) {
          sh.ncost = mn;                                                          // :1132-1133
          sh.npar = sh.uidx[sel];
        }
        __syncthreads();
      }
      const double ncost = sh.ncost;
      if (rpp::dabs(nx) > fmax || rpp::dabs(ny) > fmax) f32_ok = 0;   // outside the magnitude the margin covers
      PH(6);
      // ---------------- append :1091, rewire :1232-1246 (independent per candidate, same (theta, d) as above)
      if (tid == 0) {
        x[n] = nx;
        y[n] = ny;
        if (xf) {
          xf[n] = (float)nx;
          yf[n] = (float)ny;
        }
        cost[n] = ncost;
        parent[n] = sh.npar;
        sh.nrw = 0;
      }
      __syncthreads();
      int my_rw = 0, my_chk = 0;
      for (int e = tid; e < nu; e += TPB) {
        const double s_cost = ncost + sh.ud[e];
        if (sh.ucost[e] > s_cost) {          // near_node.cost > s_cost :1241
          my_chk++;
          if (sh.ufree[e]) {                 // check_collision(near_node, theta, d) :1244
            const int u = sh.uidx[e];
            parent[u] = n;
            cost[u] = s_cost;
            my_rw++;
          }
        }
      }
      if (my_rw) atomicAdd(&sh.nrw, my_rw);
      if (my_chk) atomicAdd(&sh.nvalid, my_chk);   // nvalid reused: rewire collision tests requested
      n++;
      __syncthreads();
      s_rw += sh.nrw;
      s_eu += sh.nvalid - nvalid;
      s_er += sh.nvalid - nvalid;
      PH(9);
      }   // eager
      // ---------------- goal bookkeeping :1094-1103
      if (tid == 0) {
        sh.flag = (rpp::py_hypot(nx - gx, ny - gy) < E) ? 1 : 0;   // is_near_goal :1226-1230 (strict)
        sh.ecoll = 0;
      }
      __syncthreads();
      if (sh.flag) {
        s_eu++;
        s_er++;
        for (int k = tid; k < c.m; k += TPB)
          if (seg_dist2(nx, ny, gx, gy, sh.ox[k], sh.oy[k]) <= sh.othr[k]) sh.ecoll = 1;   // check_segment_collision :1095
        __syncthreads();
        if (!sh.ecoll && tid == 0) {
          // get_final_course :1278-1285 + get_path_len :1193-1203 (walk first for the length, store if it improves)
          double len = 0.0, px = gx, py = gy;
          int li = n - 1, np = 1;
          while (parent[li] >= 0) {
            const double cx = x[li], cy = y[li];
            len += rpp::py_hypot(cx - px, cy - py);
            px = cx;
            py = cy;
            li = parent[li];
            np++;
          }
          len += rpp::py_hypot(sx0 - px, sy0 - py);
          np++;
          if (len < sh.cbest) {
            sh.cbest = len;
            int trunc = 0, k = 0;
            pathbuf[0] = gx;
            pathbuf[1] = gy;
            k = 1;
            li = n - 1;
            while (parent[li] >= 0) {
              if (k < c.path_cap) {
                pathbuf[2 * k] = x[li];
                pathbuf[2 * k + 1] = y[li];
              } else {
                trunc = 1;
              }
              k++;
              li = parent[li];
            }
            if (k < c.path_cap) {
              pathbuf[2 * k] = sx0;
              pathbuf[2 * k + 1] = sy0;
            } else {
              trunc = 1;
            }
            k++;
            I->path_n = k;
            I->goal_node = n - 1;
            I->status = (I->status & ~8) | 2 | (trunc ? 8 : 0);
            c.results[inst].path_cost = len;
          }
        }
        __syncthreads();
      }
    }
    if (inst == c.trace_inst && tid == 0) {
      c.tr_rx[it] = rx;
      c.tr_ry[it] = ry;
      c.tr_near[it] = ni;
      c.tr_nn[it] = nnear;
    }
    if (sh.overflow) stop = 1;
    __syncthreads();
  }

  __syncthreads();
  for (int i = tid; i < 624; i += TPB) I->rng.mt[i] = sh.rng.mt[i];
  if (tid == 0) {
    I->rng.pos = sh.rng.pos;
    I->sobol = sob;
    I->n = n;
    I->it = it;
    if (!f32_ok) I->first_goal = -3;
    PH_STORE(I);
    cbest_io[inst] = sh.cbest;
    if (it >= c.max_iter || sh.overflow) I->status |= 1;
    if (sh.overflow) I->status |= 4;
    I->iterations += s_iter;
    I->edges_unique += s_eu;
    I->edges_ref += s_er;
    I->near_hits += s_nh;
    I->near_unique += s_nu;
    I->rewires += s_rw;
    I->scan_nodes += s_sn;
    I->alg_bytes += s_ab;
    I->alg_bytes2 += s_ab2;
    I->exact_rescans += s_ex;
    c.results[inst].n_nodes = n;
    c.results[inst].status = I->status;
  }
}

}  // namespace rppi
