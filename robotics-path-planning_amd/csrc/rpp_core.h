// rpp_core.h -- scalar building blocks of the RRT / RRT* hot path, written so the
// SAME source compiles for gfx950 device code (hipcc) and for a host unit-test
// build (g++): CPython's MT19937 consumption, the reference's Sobol generator,
// CPython's math.hypot, float `**2`, steer and the per-(edge, obstacle)
// collision predicate.  No FMA contraction may be applied to this file
// (-ffp-contract=off); fused operations are written as explicit fma().
//
// Reference behaviour restated (all in /root/reference/src_path_planning/):
//   rrt_04 = 10_path_planning_01_rrt_04_rrt_star.py
//     get_random_node :1132-1139, get_random_node_sobol :1142-1153, i4_sobol :230-503,
//     steer :1086-1115, calc_distance_and_angle :1232-1238, check_collision :1216-1230,
//     check_if_outside_play_area :1204-1214
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RPP_HD __device__
#else
#define RPP_HD
#endif

#include "glibc235_fma_math.h"

namespace rpp {

// ---------------------------------------------------------------- bit helpers
RPP_HD static inline double b2d(uint64_t b) { return rpp_b2d(b); }
RPP_HD static inline uint64_t d2b(double d) { return rpp_d2b(d); }
RPP_HD static inline double dabs(double x) { return b2d(d2b(x) & 0x7fffffffffffffffULL); }
RPP_HD static inline double dinf() { return b2d(0x7ff0000000000000ULL); }

// ---------------------------------------------------------------- MT19937
// CPython's `random` (Modules/_randommodule.c): genrand_uint32, random(),
// uniform(), randint(0,100) via getrandbits(7) rejection.
struct MT {
  uint32_t mt[624];
  int32_t pos;
};

RPP_HD static inline void mt_twist(uint32_t* mt) {
  int kk;
  uint32_t y;
  for (kk = 0; kk < 624 - 397; kk++) {
    y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
    mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
  }
  for (; kk < 623; kk++) {
    y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
    mt[kk] = mt[kk - 227] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
  }
  y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
  mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
}

template <class S>
RPP_HD static inline uint32_t mt_next(S* s) {
  if (s->pos >= 624) {
    mt_twist(s->mt);
    s->pos = 0;
  }
  uint32_t y = s->mt[s->pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= (y >> 18);
  return y;
}
template <class S>
RPP_HD static inline double mt_random(S* s) {
  uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}
template <class S>
RPP_HD static inline double mt_uniform(S* s, double a, double b) {
  return a + (b - a) * mt_random(s);
}
template <class S>
RPP_HD static inline int mt_randint_0_100(S* s) {
  uint32_t r = mt_next(s) >> 25;
  while (r >= 101) r = mt_next(s) >> 25;
  return (int)r;
}
// random.seed(int) = init_by_array over the 32-bit limbs of |seed| (host side only).
static inline void mt_seed_u64(MT* s, uint64_t seed) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  int klen = key[1] ? 2 : 1;
  uint32_t* mt = s->mt;
  int i, j, k;
  mt[0] = 19650218U;
  for (i = 1; i < 624; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  i = 1;
  j = 0;
  for (k = 624; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
    i++;
    j++;
    if (i >= 624) {
      mt[0] = mt[623];
      i = 1;
    }
    if (j >= klen) j = 0;
  }
  for (k = 623; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
    i++;
    if (i >= 624) {
      mt[0] = mt[623];
      i = 1;
    }
  }
  mt[0] = 0x80000000U;
  s->pos = 624;
}

// ---------------------------------------------------------------- Sobol (dim 2)
// i4_sobol(2, index) as a planner calls it (index 0,1,2,...), rrt_04:230-503:
// direction numbers: dim 1 = 1, dim 2 from poly 3 (:360, :421-429), scaled by
// 2^(30-j) (:433-436); quasi = lastq * 2^-30 (:440, :496); Gray-code step with
// the position of the lowest zero bit of the index (:123-190, :456, :497).
struct Sobol {
  uint32_t lastq[3];
  uint32_t pad_;
  int64_t index;
};
RPP_HD static inline uint32_t sobol_v(int dim, int col) {  // col = 0..29
  if (dim == 0) return 1u << (29 - col);
  if (dim == 2) {   // third coordinate (rrt_03's i4_sobol(3, .)): poly 7, v = 1, 1, then v[j-2] ^ 2 v[j-1] ^ 4 v[j-2]
    uint32_t a = 1, b = 1;   // v[col-2], v[col-1] while stepping
    if (col < 2) return 1u << (29 - col);
    uint32_t v = 0;
    for (int j = 2; j <= col; j++) {
      v = a ^ (2u * b) ^ (4u * a);
      a = b;
      b = v;
    }
    return v << (29 - col);
  }
  uint32_t raw = 1;
  for (int j = 1; j <= col; j++) raw = raw ^ (2u * raw);
  return raw << (29 - col);
}
// i4_sobol(3, seed) with seed = 0, 1, 2, ... (rrt_03:1547)
RPP_HD static inline void sobol_next3(Sobol* s, double q[3]) {
  int l = 1;
  if (s->index == 0) {
    s->lastq[0] = s->lastq[1] = s->lastq[2] = 0;
  } else {
    int64_t n = s->index;
    while (n & 1) {
      n >>= 1;
      l++;
    }
  }
  for (int i = 0; i < 3; i++) {
    q[i] = (double)s->lastq[i] * (1.0 / 1073741824.0);
    s->lastq[i] ^= sobol_v(i, l - 1);
  }
  s->index++;
}
RPP_HD static inline void sobol_next(Sobol* s, double q[2]) {
  int l = 1;
  if (s->index == 0) {
    s->lastq[0] = s->lastq[1] = 0;
  } else {
    int64_t n = s->index;
    while (n & 1) {
      n >>= 1;
      l++;
    }
  }
  for (int i = 0; i < 2; i++) {
    q[i] = (double)s->lastq[i] * (1.0 / 1073741824.0);
    s->lastq[i] ^= sobol_v(i, l - 1);
  }
  s->index++;
}

// ---------------------------------------------------------------- math.hypot
// CPython 3.10 mathmodule.c vector_norm() for two finite arguments
// (SURVEY.md 12 A): scaled, Veltkamp-split, compensated sum of squares, one
// Newton correction.  frexp/ldexp are done on the exponent field (normal range).
RPP_HD static inline double py_hypot(double a, double b) {
  const double T27 = 134217729.0;
  double v0 = dabs(a), v1 = dabs(b);
  double mx = v0 > v1 ? v0 : v1;
  if (mx == 0.0) return 0.0;
  uint64_t be = (d2b(mx) >> 52) & 0x7ff;  // frexp exponent e = be - 1022; scale = 2^-e
  double scale = b2d((uint64_t)(2045 - be) << 52);
  double inv_scale = b2d((uint64_t)(be + 1) << 52);  // 2^e: dividing by the power of two `scale` == multiplying by it
  double csum = 1.0, f1 = 0.0, f2 = 0.0, f3 = 0.0, x, t, hi, lo, old, h;
  x = v0 * scale;
  t = x * T27; hi = t - (t - x); lo = x - hi;
  x = hi * hi; old = csum; csum += x; f1 += (old - csum) + x;
  x = 2.0 * hi * lo; old = csum; csum += x; f2 += (old - csum) + x;
  f3 += lo * lo;
  x = v1 * scale;
  t = x * T27; hi = t - (t - x); lo = x - hi;
  x = hi * hi; old = csum; csum += x; f1 += (old - csum) + x;
  x = 2.0 * hi * lo; old = csum; csum += x; f2 += (old - csum) + x;
  f3 += lo * lo;
  h = __builtin_sqrt(csum - 1.0 + (f1 + f2 + f3));
  x = h;
  t = x * T27; hi = t - (t - x); lo = x - hi;
  x = -hi * hi; old = csum; csum += x; f1 += (old - csum) + x;
  x = -2.0 * hi * lo; old = csum; csum += x; f2 += (old - csum) + x;
  x = -lo * lo; old = csum; csum += x; f3 += (old - csum) + x;
  x = csum - 1.0 + (f1 + f2 + f3);
  return (h + x / (2.0 * h)) * inv_scale;
}

// float ** 2 as CPython computes it: libm pow(|x|, 2.0) (floatobject.c float_pow);
// the glibc FMA-variant pow is < 1 ULP, not correctly rounded, so this is NOT x*x.
RPP_HD static inline double py_sq(double x) {
  if (x == 0.0) return 0.0;
  return rpp_glibc_pow(dabs(x), 2.0);
}
// (dx**2 + dy**2) exactly as rrt_04:1198 / :1335 evaluate it.
RPP_HD static inline double py_d2(double dx, double dy) { return py_sq(dx) + py_sq(dy); }
// the same with correctly rounded squares: within 2^-51 relative of py_d2 (pow is
// within 1 ULP of the true square); the scans use it as a filter only.
RPP_HD static inline double fast_d2(double dx, double dy) { return dx * dx + dy * dy; }

// ---------------------------------------------------------------- steer
// steer() of rrt_04:1086-1115 reduced to what the path needs: the polyline is
// p0 = (fx,fy), p_i = p_{i-1} + (sx,sy) for i = 1..n_expand (sequential adds,
// :1099-1101), plus the target itself when the remaining distance <= resolution
// (:1105-1110).  Collision code regenerates the points from these parameters.
struct Edge {
  double fx, fy;   // from
  double sx, sy;   // path_resolution * cos(theta), * sin(theta)
  double tx, ty;   // target
  double ex, ey;   // end point of the edge (new_node.x/.y)
  int32_t n_expand;
  int32_t snapped;
};

RPP_HD static inline void steer(Edge* e, double fx, double fy, double tx, double ty, double extend, double res) {
  double dx = tx - fx, dy = ty - fy;
  double d = py_hypot(dx, dy);
  double theta = rpp_glibc_atan2(dy, dx);  // :1232-1238
  if (extend > d) extend = d;
  int n = (int)__builtin_floor(extend / res);
  double sx = res * rpp_glibc_cos(theta), sy = res * rpp_glibc_sin(theta);
  double nx = fx, ny = fy;
  for (int i = 0; i < n; i++) {
    nx += sx;
    ny += sy;
  }
  double d2 = py_hypot(tx - nx, ty - ny);
  int snapped = d2 <= res;
  e->fx = fx; e->fy = fy; e->sx = sx; e->sy = sy; e->tx = tx; e->ty = ty;
  e->ex = snapped ? tx : nx;
  e->ey = snapped ? ty : ny;
  e->n_expand = n;
  e->snapped = snapped;
}

// check_collision (rrt_04:1216-1230) for ONE obstacle: true when some polyline
// point lies within the inflated radius, `dx*dx + dy*dy <= (size+robot_radius)**2`
// (min over points <= thr  <=>  any point <= thr).
RPP_HD static inline bool edge_hits_obstacle(const Edge& e, double ox, double oy, double thr) {
  double px = e.fx, py = e.fy;
  double dx = ox - px, dy = oy - py;
  bool hit = (dx * dx + dy * dy) <= thr;
  for (int i = 0; i < e.n_expand; i++) {
    px += e.sx;
    py += e.sy;
    dx = ox - px;
    dy = oy - py;
    hit = hit || ((dx * dx + dy * dy) <= thr);
  }
  if (e.snapped) {
    dx = ox - e.tx;
    dy = oy - e.ty;
    hit = hit || ((dx * dx + dy * dy) <= thr);
  }
  return hit;
}

// check_if_outside_play_area (rrt_04:1204-1214): true = inside / no play area.
RPP_HD static inline bool in_play_area(int has, const double* pa, double x, double y) {
  if (!has) return true;
  return !(x < pa[0] || x > pa[1] || y < pa[2] || y > pa[3]);
}

}  // namespace rpp
