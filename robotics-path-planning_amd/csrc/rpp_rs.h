// rpp_rs.h -- Reeds-Shepp steer primitive (host + device source) of rrt_06; used by rrt_rs.hip.h.
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
// one word family (:1088-1269): 1 = found (d, ct, n filled); *err set where the reference raises.
// Written as ONE code path with per-family operands instead of twelve separate bodies: on the device the twelve
// families of a steer sit in different lanes of the same wave, and every libm-grade call (pow, atan2, acos, sin,
// asin, the three mod2pi) is then executed once for all of them instead of once per family.  Each family still
// evaluates exactly the expressions of its reference function, in the reference's order:
//   0 left_straight_left :1088        1 left_straight_right :1100     2 left_x_right_x_left :1114
//   3 left_x_right_left :1127         4 left_right_x_left :1140       5 left_right_x_left_right :1154
//   6 left_x_right_left_x_right :1170 7 left_x_right90_straight_left :1189
//   8 left_x_right90_straight_right :1205   9 left_straight_right90_x_left :1218
//   10 left_straight_left90_x_right :1234   11 left_x_right90_straight_left90_x_right :1247
// (sp, cp) = sin / cos of phi: handed in by the cooperative device path, which evaluates them once per steer)
RPP_HD static inline int rs_word_sc(int w, double x, double y, double phi, double sp, double cp, double* d, char* ct,
                                    int* n, int* err) {
  const double hp = kPi / 2;
  const bool plus = (w == 1 || w == 5 || w == 6 || w == 8 || w == 10 || w == 11);   // words built on (x + sin, y - 1 - cos)
  const double zeta = plus ? x + sp : x - sp;
  const double eeta = plus ? y - 1.0 - cp : y - 1.0 + cp;
  const double u1 = py_hypot(zeta, eeta);
  const double theta = rpp_glibc_atan2(eeta, zeta);
  const bool need_sq = (w == 1 || w == 4 || w == 6 || w == 7 || w == 9 || w == 11);
  const double u1sq = need_sq ? py_sq(u1) : 0.0;
  const double u2 = (20 - u1sq) / 16;   // family 6
  bool live;
  if (w == 0) live = (0.0 <= theta && theta <= kPi);
  else if (w == 1) live = u1sq >= 4.0;
  else if (w <= 4) live = u1 <= 4.0;
  else if (w == 5) live = u1 <= 2;
  else if (w == 6) live = (0 <= u2 && u2 <= 1);
  else if (w <= 10) live = u1 >= 2.0;
  else live = u1 >= 4.0;
  if (!live) return 0;
  // sqrt(u1 ** 2 - 4) and the atan2 built on it (families 1, 7, 9, 11)
  const bool need_r = (w == 1 || w == 7 || w == 9 || w == 11);
  const double r = need_r ? __builtin_sqrt(u1sq - 4.0) : 0.0;
  double A = 0.0;
  if (need_r) A = rpp_glibc_atan2(w == 9 ? r : 2.0, w == 9 ? 2.0 : r);
  // acos (families 2 .. 6)
  double AC = 0.0;
  if (w >= 2 && w <= 6) {
    const double arg = (w <= 3) ? 0.25 * u1 : (w == 4 ? 1 - u1sq * 0.125 : (w == 5 ? (u1 + 2) * 0.25 : u2));
    AC = rpp_glibc_acos(arg);
  }
  // asin(2 sin(u) / u1) (families 4, 6)
  double A2 = 0.0;
  if (w == 4 || w == 6) {
    const double num = 2 * rpp_glibc_sin(AC);
    if (u1 == 0.0) { *err = -3; return 0; }
    const double q = num / u1;
    if (!(dabs(q) <= 1.0)) { *err = -4; return 0; }
    A2 = rpp_glibc_asin(q);
  }
  // t
  double targ;
  switch (w) {
    case 0: targ = 0.0; break;
    case 1: targ = theta + A; break;
    case 2: case 3: targ = AC + theta + hp; break;
    case 4: targ = -A2 + theta + hp; break;
    case 5: targ = theta + AC + hp; break;
    case 6: targ = theta + A2 + hp; break;
    case 7: case 11: targ = theta + A + hp; break;
    case 8: targ = theta + hp; break;
    case 9: targ = theta - A + hp; break;
    default: targ = theta; break;   // 10
  }
  const double t = (w == 0) ? theta : rs_mod2pi(targ);
  // u where it is an angle of its own (families 2, 3, 5)
  double u = 0.0;
  if (w == 2 || w == 3 || w == 5) u = rs_mod2pi(w == 5 ? AC : kPi - 2 * AC);
  // v
  double varg;
  switch (w) {
    case 0: varg = phi - theta; break;
    case 1: case 6: case 11: varg = t - phi; break;
    case 2: varg = phi - t - u; break;
    case 3: varg = -phi + t + u; break;
    case 4: varg = t - AC - phi; break;
    case 5: varg = phi - t + 2 * u; break;
    case 7: varg = t - phi + hp; break;
    case 9: varg = t - phi - hp; break;
    default: varg = phi - t - hp; break;   // 8, 10
  }
  const double v = rs_mod2pi(varg);
  switch (w) {
    case 0:
      if (0.0 <= v && v <= kPi) { d[0] = theta; d[1] = u1; d[2] = v; rs_ct(ct, "LSL"); *n = 3; return 1; }
      return 0;
    case 1:
      if (t >= 0.0 && v >= 0.0) { d[0] = t; d[1] = r; d[2] = v; rs_ct(ct, "LSR"); *n = 3; return 1; }
      return 0;
    case 2: d[0] = t; d[1] = -u; d[2] = v; rs_ct(ct, "LRL"); *n = 3; return 1;
    case 3: d[0] = t; d[1] = -u; d[2] = -v; rs_ct(ct, "LRL"); *n = 3; return 1;
    case 4: d[0] = t; d[1] = AC; d[2] = -v; rs_ct(ct, "LRL"); *n = 3; return 1;
    case 5:
      if (t >= 0 && u >= 0 && v >= 0) { d[0] = t; d[1] = u; d[2] = -u; d[3] = -v; rs_ct(ct, "LRLR"); *n = 4; return 1; }
      return 0;
    case 6:
      if (t >= 0 && v >= 0) { d[0] = t; d[1] = -AC; d[2] = -AC; d[3] = v; rs_ct(ct, "LRLR"); *n = 4; return 1; }
      return 0;
    case 7:
      if (t >= 0 && v >= 0) { d[0] = t; d[1] = -hp; d[2] = -(r - 2); d[3] = -v; rs_ct(ct, "LRSL"); *n = 4; return 1; }
      return 0;
    case 8:
      if (t >= 0 && v >= 0) { d[0] = t; d[1] = -hp; d[2] = -(u1 - 2); d[3] = -v; rs_ct(ct, "LRSR"); *n = 4; return 1; }
      return 0;
    case 9:
      if (t >= 0 && v >= 0) { d[0] = t; d[1] = r - 2; d[2] = hp; d[3] = -v; rs_ct(ct, "LSRL"); *n = 4; return 1; }
      return 0;
    case 10:
      if (t >= 0 && v >= 0) { d[0] = t; d[1] = u1 - 2; d[2] = hp; d[3] = -v; rs_ct(ct, "LSLR"); *n = 4; return 1; }
      return 0;
    default:
      if (t >= 0 && v >= 0) {
        d[0] = t; d[1] = -hp; d[2] = -(r - 4); d[3] = -hp; d[4] = v; rs_ct(ct, "LRSLR"); *n = 5; return 1;
      }
      return 0;
  }
}
RPP_HD static inline int rs_word(int w, double x, double y, double phi, double* d, char* ct, int* n, int* err) {
  return rs_word_sc(w, x, y, phi, rpp_glibc_sin(phi), rpp_glibc_cos(phi), d, ct, n, err);
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
// _interpolate :1379-1401, split so that what is constant along a segment is computed once: (cs, sn) = cos / sin of
// the segment origin's yaw (straight) or of its negative (arcs); (dx, dy) = the displacement the reference adds
// to the origin
// (the arcs' cos(-yaw), sin(-yaw) are taken as cos(yaw), -sin(yaw): the replicas are even / odd bit for bit, which
// tests/native/core_host_check.cpp pins; one code path for straight and arc lanes on the device)
RPP_HD static inline void rs_seg_trig(char mode, double oyaw, double* cs, double* sn) {
  const double c0 = rpp_glibc_cos(oyaw), s0 = rpp_glibc_sin(oyaw);
  *cs = c0;
  *sn = (mode == 'S') ? s0 : -s0;
}
// (sd, cd) = sin / cos of dist (unused for straight segments)
RPP_HD static inline void rs_delta_sc(double dist, char mode, double maxc, double cs, double sn, double sd, double cd,
                                      double* dx, double* dy) {
  if (mode == 'S') {
    *dx = dist / maxc * cs;
    *dy = dist / maxc * sn;
  } else {
    const double ldx = sd / maxc;
    const double ldy = (mode == 'L') ? (1.0 - cd) / maxc : (1.0 - cd) / -maxc;
    *dx = cs * ldx + sn * ldy;
    *dy = -sn * ldx + cs * ldy;
  }
}
RPP_HD static inline void rs_delta(double dist, char mode, double maxc, double cs, double sn, double* dx, double* dy) {
  if (mode == 'S')
    rs_delta_sc(dist, mode, maxc, cs, sn, 0.0, 0.0, dx, dy);
  else
    rs_delta_sc(dist, mode, maxc, cs, sn, rpp_glibc_sin(dist), rpp_glibc_cos(dist), dx, dy);
}
RPP_HD static inline double rs_yaw_after(double dist, char mode, double oyaw) {
  return (mode == 'S') ? oyaw : ((mode == 'L') ? oyaw + dist : oyaw - dist);
}
RPP_HD static inline void rs_interp(double dist, char mode, double maxc, double ox, double oy, double oyaw, double* x,
                                    double* y, double* yaw) {
  double cs, sn, dx, dy;
  rs_seg_trig(mode, oyaw, &cs, &sn);
  rs_delta(dist, mode, maxc, cs, sn, &dx, &dy);
  *x = ox + dx;
  *y = oy + dy;
  *yaw = rs_yaw_after(dist, mode, oyaw);
}
// ---- the solver in separable parts (the iteration kernel of rrt_06 runs them cooperatively; rs_plan below is the
// same sequence on one thread, and is what the known-answer vectors pin) ----------------------------------------
// local frame of generate_path :1286-1296
struct RsFrame {
  double x, y, dth, step;
  double c, s;   // cos / sin of the start yaw
};
// (c, s) = cos / sin of the start yaw
RPP_HD static inline void rs_frame_sc(double sx, double sy, double syaw, double gx, double gy, double gyaw, double maxc,
                                      double step_size, double c, double s, RsFrame* F) {
  const double dx = gx - sx, dy = gy - sy;
  F->dth = gyaw - syaw;
  F->x = (c * dx + s * dy) * maxc;
  F->y = (-s * dx + c * dy) * maxc;
  F->c = c;
  F->s = s;
  F->step = step_size * maxc;
}
RPP_HD static inline void rs_frame(double sx, double sy, double syaw, double gx, double gy, double gyaw, double maxc,
                                   double step_size, RsFrame* F) {
  rs_frame_sc(sx, sy, syaw, gx, gy, gyaw, maxc, step_size, rpp_glibc_cos(syaw), rpp_glibc_sin(syaw), F);
}
// One (word family w, symmetry var) of generate_path :1298-1340.  Returns 0: no path from this variant, 1: a path
// (d, ct, n filled, flips applied), 2: "Step size too large" (the reference returns [] there and then, :1071-1074),
// < 0: the reference raises.
// (sdth, cdth) = sin / cos of F.dth; the variants with phi = -dth take -sin, cos (the replicas are odd / even bit for
// bit: tests/native/core_host_check.cpp)
RPP_HD static inline int rs_variant_sc(int w, int var, const RsFrame& F, double sdth, double cdth, double* d, char* ct, int* n);
RPP_HD static inline int rs_variant(int w, int var, const RsFrame& F, double* d, char* ct, int* n) {
  return rs_variant_sc(w, var, F, rpp_glibc_sin(F.dth), rpp_glibc_cos(F.dth), d, ct, n);
}
RPP_HD static inline int rs_variant_sc(int w, int var, const RsFrame& F, double sdth, double cdth, double* d, char* ct, int* n) {
  const double xx = (var == 1 || var == 3) ? -F.x : F.x, yy = (var >= 2) ? -F.y : F.y;
  const bool neg = (var == 1 || var == 2);
  const double pp = neg ? -F.dth : F.dth;
  int err = 0;
  *n = 0;
  if (!rs_word_sc(w, xx, yy, pp, neg ? -sdth : sdth, cdth, d, ct, n, &err)) return err ? err : 0;
  const double tot = rs_sum_abs(d, *n);
  for (int i = 0; i < *n; i++)
    if (0.1 * tot < dabs(d[i]) && dabs(d[i]) < F.step) return 2;
  if (var == 1 || var == 3)
    for (int i = 0; i < *n; i++) d[i] = -d[i];                  // timeflip
  if (var >= 2)
    for (int i = 0; ct[i]; i++) ct[i] = (ct[i] == 'L') ? 'R' : (ct[i] == 'R' ? 'L' : 'S');   // reflect
  return 1;
}
// the word's letters packed two bits each (equal codes <=> equal ctypes lists)
RPP_HD static inline uint32_t rs_code(const char* ct) {
  uint32_t c = 0;
  for (int i = 0; ct[i]; i++) c |= (uint32_t)(ct[i] == 'L' ? 1 : (ct[i] == 'S' ? 2 : 3)) << (2 * i);
  return c;
}
// set_path :1061-1080 over the variants in the reference's order + `paths.index(min(...))` :1436.
// st[k], d[k][5], ct[k][6], n[k] for k = 4 * w + var.  Returns the chosen k, -1: None, < -1: the reference raises.
template <typename D5, typename C6>
RPP_HD static inline int rs_select(const int32_t* st, const D5* d, const C6* ct, const int32_t* n, double step,
                                   double maxc) {
  int kept[RS_MAXP];
  double keptL[RS_MAXP];
  int np = 0;
  for (int k = 0; k < 48; k++) {
    if (st[k] == 0) continue;
    if (st[k] < 0) return st[k];   // -3 / -4
    if (st[k] == 2) return -1;
    const double L = rs_sum_abs(d[k], n[k]);
    bool skip = false;
    for (int i = 0; i < np; i++)
      if (rs_same(ct[kept[i]], ct[k]) && (keptL[i] - L) <= step) skip = true;
    if (skip || L <= step || np >= RS_MAXP) continue;
    kept[np] = k;
    keptL[np] = L;
    np++;
  }
  if (np == 0) return -1;
  int bi = 0;
  double bl = dabs(keptL[0] / maxc);
  for (int i = 1; i < np; i++) {
    const double l = dabs(keptL[i] / maxc);
    if (l < bl) {
      bl = l;
      bi = i;
    }
  }
  return kept[bi];
}
// generate_local_course :1355-1377 prepared for random access: segment origins, np.arange counts
struct RsCourse {
  double len[5], ddist[5];
  double ox[5], oy[5], oyaw[5];
  double cs[5], sn[5];   // rs_seg_trig of the segment
  int32_t cnt[5];      // arange points of the segment; the segment has cnt + 1 points (np.append(.., length))
  int32_t first[6];    // index of the segment's first point in the polyline
  char ct[6];
  int32_t nl, total;
  double sx, sy, syaw, cg, sg, maxc;
};
RPP_HD static inline void rs_course(const double* len, const char* ct, int nl, double sx, double sy, double syaw,
                                    double maxc, double step_size, RsCourse* C) {
  C->cg = rpp_glibc_cos(-syaw);
  C->sg = rpp_glibc_sin(-syaw);
  C->sx = sx;
  C->sy = sy;
  C->syaw = syaw;
  C->maxc = maxc;
  C->nl = nl;
  double ox = 0.0, oy = 0.0, oyaw = 0.0;
  const double ds = step_size * maxc;
  int tot = 0;
  for (int sgm = 0; sgm < nl; sgm++) {
    const double length = len[sgm];
    const double d_dist = length >= 0.0 ? ds : -ds;
    const double q = (length - 0.0) / d_dist;            // np.arange(0.0, length, d_dist)
    const long cnt = (q > 0.0) ? (long)__builtin_ceil(q) : 0;
    C->len[sgm] = length;
    C->ddist[sgm] = d_dist;
    C->ct[sgm] = ct[sgm];
    C->ox[sgm] = ox;
    C->oy[sgm] = oy;
    C->oyaw[sgm] = oyaw;
    C->cnt[sgm] = (int32_t)cnt;
    C->first[sgm] = tot;
    tot += (int32_t)cnt + 1;
    double dx, dy;
    rs_seg_trig(ct[sgm], oyaw, &C->cs[sgm], &C->sn[sgm]);
    rs_delta(length, ct[sgm], maxc, C->cs[sgm], C->sn[sgm], &dx, &dy);   // the segment's last point = next origin
    ox = ox + dx;
    oy = oy + dy;
    oyaw = rs_yaw_after(length, ct[sgm], oyaw);
  }
  C->ct[nl] = 0;
  C->first[nl] = tot;
  C->total = tot;
}
// world-frame point k of the course (:1411-1417)
RPP_HD static inline void rs_point(const RsCourse& C, int k, double* wx, double* wy, double* wyaw) {
  int sgm = 0;
  while (sgm + 1 < C.nl && k >= C.first[sgm + 1]) sgm++;
  const int i = k - C.first[sgm];
  const double dist = (i < C.cnt[sgm]) ? 0.0 + (double)i * C.ddist[sgm] : C.len[sgm];
  double dx, dy;
  rs_delta(dist, C.ct[sgm], C.maxc, C.cs[sgm], C.sn[sgm], &dx, &dy);
  const double lx = C.ox[sgm] + dx, ly = C.oy[sgm] + dy, lyaw = rs_yaw_after(dist, C.ct[sgm], C.oyaw[sgm]);
  *wx = C.cg * lx + C.sg * ly + C.sx;
  *wy = -C.sg * lx + C.cg * ly + C.sy;
  *wyaw = angle_mod_pi(lyaw + C.syaw);
}
// reeds_shepp_path_planning :1426-1441; writes at most cap points
RPP_HD static inline void rs_plan(double sx, double sy, double syaw, double gx, double gy, double gyaw, double maxc,
                                  double step_size, double* px, double* py, double* pyaw, int cap, RsResult* R) {
  R->n = 0;
  R->err = 0;
  R->nl = 0;
  RsFrame F;
  rs_frame(sx, sy, syaw, gx, gy, gyaw, maxc, step_size, &F);
  int32_t st[48], n[48];
  double d[48][5];
  char ct[48][6];
  for (int k = 0; k < 48; k++) {
    int nn = 0;
    st[k] = rs_variant(k >> 2, k & 3, F, d[k], ct[k], &nn);
    n[k] = nn;
    if (st[k] < 0 || st[k] == 2) {   // the reference stops here; nothing later is looked at
      for (int q = k + 1; q < 48; q++) st[q] = 0;
      break;
    }
  }
  const int sel = rs_select(st, d, ct, n, F.step, maxc);
  if (sel < -1) {
    R->err = sel;
    return;
  }
  if (sel < 0) return;
  RsCourse C;
  rs_course(d[sel], ct[sel], n[sel], sx, sy, syaw, maxc, step_size, &C);
  for (int k = 0; k < C.total && k < cap; k++) rs_point(C, k, &px[k], &py[k], &pyaw[k]);
  for (int i = 0; i < n[sel]; i++) R->len[i] = d[sel][i] / maxc;
  rs_ct(R->ct, ct[sel]);
  R->nl = n[sel];
  R->n = C.total;
}

}  // namespace rpp
