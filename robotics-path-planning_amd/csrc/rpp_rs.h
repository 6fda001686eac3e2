// rpp_rs.h -- Reeds-Shepp steer primitive (host + device source): groundwork for rrt_06, not yet used by a kernel.
// Reference: 10_path_planning_01_rrt_06_rrt_star_reeds_shepp_path.py -- reeds_shepp_path_planning :1426-1441,
// calc_paths :1403-1424, generate_path :1286-1342 (12 word families x {identity, timeflip, reflect, both}),
// set_path :1061-1080, mod2pi :1051-1059, polar :1082-1085, the word functions :1088-1269,
// calc_interpolate_dists_list :1344-1353 (np.arange), generate_local_course :1355-1377, interpolate :1379-1401.
// Same arithmetic as the reference operation by operation (glibc replicas, CPython hypot / **2, numpy mod and
// arange forms); pinned on the host against tests/golden/rs_kat.npz (tests/native/rs_host_check.cpp).
#pragma once
#include "rpp_dubins.h"

namespace rpp {

constexpr int RS_MAXP = 48;   // candidate paths kept by set_path (12 words x 4 variants)
struct RsCand {
  double len[5];
  char ct[6];
  int32_t nl;
  double L;
};
struct RsResult {
  double len[5];     // segment lengths of the chosen path, already divided by the curvature (:1420)
  char ct[6];
  int32_t nl;
  int32_t n;         // points written; 0: the reference returns None
  int32_t err;       // 0, -3: the reference raises ZeroDivisionError, -4: ValueError
};

RPP_HD static inline double rs_copysign(double mag, double sgn) {
  return b2d((d2b(mag) & 0x7fffffffffffffffULL) | (d2b(sgn) & 0x8000000000000000ULL));
}
RPP_HD static inline double rs_mod2pi(double x) {   // :1051-1059
  double v = np_mod(x, rs_copysign(2.0 * kPi, x));
  if (v < -kPi)
    v += 2.0 * kPi;
  else if (v > kPi)
    v -= 2.0 * kPi;
  return v;
}
RPP_HD static inline void rs_ct(char* ct, const char* w) {
  int i = 0;
  for (; w[i]; i++) ct[i] = w[i];
  ct[i] = 0;
}
// one word family (:1088-1269): 1 = found (d, ct, n filled); *err set where the reference raises
RPP_HD static inline int rs_word(int w, double x, double y, double phi, double* d, char* ct, int* n, int* err) {
  const double sp = rpp_glibc_sin(phi), cp = rpp_glibc_cos(phi);
  const bool plus = (w == 1 || w == 5 || w == 6 || w == 8 || w == 10 || w == 11);   // words built on (x + sin, y - 1 - cos)
  const double zeta = plus ? x + sp : x - sp;
  const double eeta = plus ? y - 1.0 - cp : y - 1.0 + cp;
  double u1 = py_hypot(zeta, eeta);
  const double theta = rpp_glibc_atan2(eeta, zeta);
  double u, t, v, A;
  switch (w) {
    case 0:   // left_straight_left
      if (0.0 <= theta && theta <= kPi) {
        v = rs_mod2pi(phi - theta);
        if (0.0 <= v && v <= kPi) { d[0] = theta; d[1] = u1; d[2] = v; rs_ct(ct, "LSL"); *n = 3; return 1; }
      }
      return 0;
    case 1:   // left_straight_right
      u1 = py_sq(u1);
      if (u1 >= 4.0) {
        u = __builtin_sqrt(u1 - 4.0);
        t = rs_mod2pi(theta + rpp_glibc_atan2(2.0, u));
        v = rs_mod2pi(t - phi);
        if (t >= 0.0 && v >= 0.0) { d[0] = t; d[1] = u; d[2] = v; rs_ct(ct, "LSR"); *n = 3; return 1; }
      }
      return 0;
    case 2:   // left_x_right_x_left
      if (u1 <= 4.0) {
        A = rpp_glibc_acos(0.25 * u1);
        t = rs_mod2pi(A + theta + kPi / 2);
        u = rs_mod2pi(kPi - 2 * A);
        v = rs_mod2pi(phi - t - u);
        d[0] = t; d[1] = -u; d[2] = v; rs_ct(ct, "LRL"); *n = 3; return 1;
      }
      return 0;
    case 3:   // left_x_right_left
      if (u1 <= 4.0) {
        A = rpp_glibc_acos(0.25 * u1);
        t = rs_mod2pi(A + theta + kPi / 2);
        u = rs_mod2pi(kPi - 2 * A);
        v = rs_mod2pi(-phi + t + u);
        d[0] = t; d[1] = -u; d[2] = -v; rs_ct(ct, "LRL"); *n = 3; return 1;
      }
      return 0;
    case 4:   // left_right_x_left
      if (u1 <= 4.0) {
        u = rpp_glibc_acos(1 - py_sq(u1) * 0.125);
        const double num = 2 * rpp_glibc_sin(u);
        if (u1 == 0.0) { *err = -3; return 0; }
        const double q = num / u1;
        if (!(dabs(q) <= 1.0)) { *err = -4; return 0; }
        A = rpp_glibc_asin(q);
        t = rs_mod2pi(-A + theta + kPi / 2);
        v = rs_mod2pi(t - u - phi);
        d[0] = t; d[1] = u; d[2] = -v; rs_ct(ct, "LRL"); *n = 3; return 1;
      }
      return 0;
    case 5:   // left_right_x_left_right
      if (u1 <= 2) {
        A = rpp_glibc_acos((u1 + 2) * 0.25);
        t = rs_mod2pi(theta + A + kPi / 2);
        u = rs_mod2pi(A);
        v = rs_mod2pi(phi - t + 2 * u);
        if (t >= 0 && u >= 0 && v >= 0) { d[0] = t; d[1] = u; d[2] = -u; d[3] = -v; rs_ct(ct, "LRLR"); *n = 4; return 1; }
      }
      return 0;
    case 6: {   // left_x_right_left_x_right
      const double u2 = (20 - py_sq(u1)) / 16;
      if (0 <= u2 && u2 <= 1) {
        u = rpp_glibc_acos(u2);
        const double num = 2 * rpp_glibc_sin(u);
        if (u1 == 0.0) { *err = -3; return 0; }
        const double q = num / u1;
        if (!(dabs(q) <= 1.0)) { *err = -4; return 0; }
        A = rpp_glibc_asin(q);
        t = rs_mod2pi(theta + A + kPi / 2);
        v = rs_mod2pi(t - phi);
        if (t >= 0 && v >= 0) { d[0] = t; d[1] = -u; d[2] = -u; d[3] = v; rs_ct(ct, "LRLR"); *n = 4; return 1; }
      }
      return 0;
    }
    case 7:   // left_x_right90_straight_left
      if (u1 >= 2.0) {
        const double r = __builtin_sqrt(py_sq(u1) - 4);
        u = r - 2;
        A = rpp_glibc_atan2(2.0, r);
        t = rs_mod2pi(theta + A + kPi / 2);
        v = rs_mod2pi(t - phi + kPi / 2);
        if (t >= 0 && v >= 0) { d[0] = t; d[1] = -kPi / 2; d[2] = -u; d[3] = -v; rs_ct(ct, "LRSL"); *n = 4; return 1; }
      }
      return 0;
    case 8:   // left_x_right90_straight_right
      if (u1 >= 2.0) {
        t = rs_mod2pi(theta + kPi / 2);
        u = u1 - 2;
        v = rs_mod2pi(phi - t - kPi / 2);
        if (t >= 0 && v >= 0) { d[0] = t; d[1] = -kPi / 2; d[2] = -u; d[3] = -v; rs_ct(ct, "LRSR"); *n = 4; return 1; }
      }
      return 0;
    case 9:   // left_straight_right90_x_left
      if (u1 >= 2.0) {
        const double r = __builtin_sqrt(py_sq(u1) - 4);
        u = r - 2;
        A = rpp_glibc_atan2(r, 2.0);
        t = rs_mod2pi(theta - A + kPi / 2);
        v = rs_mod2pi(t - phi - kPi / 2);
        if (t >= 0 && v >= 0) { d[0] = t; d[1] = u; d[2] = kPi / 2; d[3] = -v; rs_ct(ct, "LSRL"); *n = 4; return 1; }
      }
      return 0;
    case 10:   // left_straight_left90_x_right
      if (u1 >= 2.0) {
        t = rs_mod2pi(theta);
        u = u1 - 2;
        v = rs_mod2pi(phi - t - kPi / 2);
        if (t >= 0 && v >= 0) { d[0] = t; d[1] = u; d[2] = kPi / 2; d[3] = -v; rs_ct(ct, "LSLR"); *n = 4; return 1; }
      }
      return 0;
    default:   // left_x_right90_straight_left90_x_right
      if (u1 >= 4.0) {
        const double r = __builtin_sqrt(py_sq(u1) - 4);
        u = r - 4;
        A = rpp_glibc_atan2(2.0, r);
        t = rs_mod2pi(theta + A + kPi / 2);
        v = rs_mod2pi(t - phi);
        if (t >= 0 && v >= 0) {
          d[0] = t; d[1] = -kPi / 2; d[2] = -u; d[3] = -kPi / 2; d[4] = v; rs_ct(ct, "LRSLR"); *n = 5; return 1;
        }
      }
      return 0;
  }
}
RPP_HD static inline double rs_sum_abs(const double* d, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += dabs(d[i]);
  return s;
}
RPP_HD static inline bool rs_same(const char* a, const char* b) {
  for (int i = 0;; i++) {
    if (a[i] != b[i]) return false;
    if (!a[i]) return true;
  }
}
// _interpolate :1379-1401
RPP_HD static inline void rs_interp(double dist, char mode, double maxc, double ox, double oy, double oyaw, double* x,
                                    double* y, double* yaw) {
  if (mode == 'S') {
    *x = ox + dist / maxc * rpp_glibc_cos(oyaw);
    *y = oy + dist / maxc * rpp_glibc_sin(oyaw);
    *yaw = oyaw;
  } else {
    const double ldx = rpp_glibc_sin(dist) / maxc;
    const double ldy = (mode == 'L') ? (1.0 - rpp_glibc_cos(dist)) / maxc : (1.0 - rpp_glibc_cos(dist)) / -maxc;
    const double c = rpp_glibc_cos(-oyaw), s = rpp_glibc_sin(-oyaw);
    *x = ox + (c * ldx + s * ldy);
    *y = oy + (-s * ldx + c * ldy);
    *yaw = (mode == 'L') ? oyaw + dist : oyaw - dist;
  }
}
// reeds_shepp_path_planning :1426-1441; writes at most cap points
RPP_HD static inline void rs_plan(double sx, double sy, double syaw, double gx, double gy, double gyaw, double maxc,
                                  double step_size, double* px, double* py, double* pyaw, int cap, RsResult* R) {
  RsCand cand[RS_MAXP];
  int np = 0;
  R->n = 0;
  R->err = 0;
  R->nl = 0;
  // generate_path :1286-1342
  const double dx = gx - sx, dy = gy - sy, dth = gyaw - syaw;
  const double c = rpp_glibc_cos(syaw), s = rpp_glibc_sin(syaw);
  const double x = (c * dx + s * dy) * maxc, y = (-s * dx + c * dy) * maxc;
  const double step = step_size * maxc;
  for (int w = 0; w < 12; w++) {
    for (int var = 0; var < 4; var++) {
      const double xx = (var == 1 || var == 3) ? -x : x, yy = (var >= 2) ? -y : y;
      const double pp = (var == 1 || var == 2) ? -dth : dth;
      double d[5];
      char ct[6];
      int n = 0, err = 0;
      if (!rs_word(w, xx, yy, pp, d, ct, &n, &err)) {
        if (err) {
          R->err = err;
          return;
        }
        continue;
      }
      const double tot = rs_sum_abs(d, n);
      for (int i = 0; i < n; i++)
        if (0.1 * tot < dabs(d[i]) && dabs(d[i]) < step) return;   // "Step size too large" -> [] -> None
      if (var == 1 || var == 3)
        for (int i = 0; i < n; i++) d[i] = -d[i];                  // timeflip
      if (var >= 2)
        for (int i = 0; ct[i]; i++) ct[i] = (ct[i] == 'L') ? 'R' : (ct[i] == 'R' ? 'L' : 'S');   // reflect
      // set_path :1061-1080
      const double L = rs_sum_abs(d, n);
      bool skip = false;
      for (int i = 0; i < np; i++)
        if (rs_same(cand[i].ct, ct) && (rs_sum_abs(cand[i].len, cand[i].nl) - L) <= step) skip = true;
      if (skip || L <= step || np >= RS_MAXP) continue;
      for (int i = 0; i < n; i++) cand[np].len[i] = d[i];
      rs_ct(cand[np].ct, ct);
      cand[np].nl = n;
      cand[np].L = L;
      np++;
    }
  }
  if (np == 0) return;
  int bi = 0;
  double bl = dabs(cand[0].L / maxc);
  for (int i = 1; i < np; i++) {   // paths.index(min(paths, key=abs(L))) :1436
    const double l = dabs(cand[i].L / maxc);
    if (l < bl) {
      bl = l;
      bi = i;
    }
  }
  const RsCand& P = cand[bi];
  // generate_local_course :1355-1377 + global conversion :1411-1417
  const double cg = rpp_glibc_cos(-syaw), sg = rpp_glibc_sin(-syaw);
  double ox = 0.0, oy = 0.0, oyaw = 0.0;
  const double ds = step_size * maxc;
  int n = 0;
  for (int sgm = 0; sgm < P.nl; sgm++) {
    const double length = P.len[sgm];
    const char md = P.ct[sgm];
    const double d_dist = length >= 0.0 ? ds : -ds;
    const double q = (length - 0.0) / d_dist;            // np.arange(0.0, length, d_dist)
    const long cnt = (q > 0.0) ? (long)__builtin_ceil(q) : 0;
    double lx = ox, ly = oy, lyaw = oyaw;
    for (long i = 0; i <= cnt; i++) {
      const double dist = (i < cnt) ? 0.0 + (double)i * d_dist : length;   // np.append(interp_dists, length)
      rs_interp(dist, md, maxc, ox, oy, oyaw, &lx, &ly, &lyaw);
      if (n < cap) {
        px[n] = cg * lx + sg * ly + sx;
        py[n] = -sg * lx + cg * ly + sy;
        pyaw[n] = angle_mod_pi(lyaw + syaw);
      }
      n++;
    }
    ox = lx;
    oy = ly;
    oyaw = lyaw;
  }
  for (int i = 0; i < P.nl; i++) R->len[i] = P.len[i] / maxc;
  rs_ct(R->ct, P.ct);
  R->nl = P.nl;
  R->n = n;
}

}  // namespace rpp
