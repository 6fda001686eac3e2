// rpp_dubins.h -- scalar Dubins-path building blocks (host + device source, like rpp_core.h).
// Reference: /root/reference/src_path_planning/10_path_planning_01_rrt_05_rrt_star_dubins_path.py (rrt_05)
//   rot_mat_2d :935-952 (scipy Rotation.from_euler('z', a).as_matrix()), angle_mod :956-1012 (numpy `%`),
//   plan_dubins_path :1021-1109, _LSL.._LRL :1125-1198, _dubins_path_planning_from_origin :1201-1229,
//   _interpolate :1232-1255, _generate_local_course :1258-1278.
// Third-party arithmetic restated, pinned by tests/golden/dubins_kat.npz:
//   scipy: m = [[w2 - z2, -2zw], [2zw, w2 - z2]], z = sin(a/2), w = cos(a/2);
//   numpy (2,)@(2,2) and (n,2)@(2,2): out_j = fma(p1, m[1][j], p0 * m[0][j]);
//   numpy float `%`: fmod (exact) + sign fix.
#pragma once
#include "rpp_core.h"

namespace rpp {

// C fmod: exact remainder with the sign of x (bit-serial long division on the significands)
RPP_HD static inline double fmod_exact(double x, double y) {
  uint64_t ux = d2b(x), uy = d2b(y);
  int ex = (int)((ux >> 52) & 0x7ff), ey = (int)((uy >> 52) & 0x7ff);
  const uint64_t sx = ux >> 63;
  if ((uy << 1) == 0 || ey == 0x7ff || ex == 0x7ff) return (x * y) / (x * y);
  if ((ux << 1) <= (uy << 1)) {
    if ((ux << 1) == (uy << 1)) return 0.0 * x;
    return x;
  }
  uint64_t i;
  if (!ex) {
    for (i = ux << 12; (i >> 63) == 0; ex--, i <<= 1) {
    }
    ux <<= -ex + 1;
  } else {
    ux &= ~0ULL >> 12;
    ux |= 1ULL << 52;
  }
  if (!ey) {
    for (i = uy << 12; (i >> 63) == 0; ey--, i <<= 1) {
    }
    uy <<= -ey + 1;
  } else {
    uy &= ~0ULL >> 12;
    uy |= 1ULL << 52;
  }
  for (; ex > ey; ex--) {
    i = ux - uy;
    if ((i >> 63) == 0) {
      if (i == 0) return 0.0 * x;
      ux = i;
    }
    ux <<= 1;
  }
  i = ux - uy;
  if ((i >> 63) == 0) {
    if (i == 0) return 0.0 * x;
    ux = i;
  }
  for (; (ux >> 52) == 0; ux <<= 1, ex--) {
  }
  if (ex > 0) {
    ux -= 1ULL << 52;
    ux |= (uint64_t)ex << 52;
  } else {
    ux >>= -ex + 1;
  }
  ux |= sx << 63;
  return b2d(ux);
}

// numpy float remainder (npy_remainder): Python-style floored modulo
RPP_HD static inline double np_mod(double a, double b) {
  double m = fmod_exact(a, b);
  if (b == 0.0) return m;
  if (m != 0.0) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = b2d(d2b(b) & 0x8000000000000000ULL);  // copysign(0, b)
  }
  return m;
}
constexpr double kPi = 3.141592653589793;
constexpr double k2Pi = 6.283185307179586;
RPP_HD static inline double mod2pi(double t) { return np_mod(t, k2Pi); }                 // angle_mod(zero_2_2pi=True)
RPP_HD static inline double angle_mod_pi(double x) { return np_mod(x + kPi, k2Pi) - kPi; }   // default branch :1000

RPP_HD static inline void rot_mat_2d(double a, double m[4]) {   // :952
  const double z = rpp_glibc_sin(a / 2), w = rpp_glibc_cos(a / 2);
  const double z2 = z * z, w2 = w * w, zw = z * w;
  m[0] = (0.0 - 0.0 - z2) + w2;
  m[1] = 2 * (0.0 - zw);
  m[2] = 2 * (0.0 + zw);
  m[3] = (-0.0 + 0.0 - z2) + w2;
}

// rot_mat_2d(a) and rot_mat_2d(-a) from one sin/cos pair: (-a)/2 == -(a/2), sin is odd and cos even bit for bit
// (glibc's __sin works on |x| and restores the sign; checked on 2e7 arguments), so z -> -z, w -> w in :952.
RPP_HD static inline void rot_mat_2d_pair(double a, double m[4], double mb[4]) {
  const double z = rpp_glibc_sin(a / 2), w = rpp_glibc_cos(a / 2);
  const double z2 = z * z, w2 = w * w, zw = z * w, nzw = -zw;
  m[0] = (0.0 - 0.0 - z2) + w2;
  m[1] = 2 * (0.0 - zw);
  m[2] = 2 * (0.0 + zw);
  m[3] = (-0.0 + 0.0 - z2) + w2;
  mb[0] = (0.0 - 0.0 - z2) + w2;
  mb[1] = 2 * (0.0 - nzw);
  mb[2] = 2 * (0.0 + nzw);
  mb[3] = (-0.0 + 0.0 - z2) + w2;
}

// one Dubins word (wi in _PATH_TYPE_MAP order LSL,RSR,LSR,RSL,RLR,LRL :1797); trig = sin a, sin b, cos a, cos b, cos(a-b)
// d2 = d ** 2; a1 = atan2(cb - ca, d + sa - sb) (LSL), a2 = atan2(ca - cb, d - sa + sb) (RSR and RLR evaluate the same
// call), a3 = atan2(ca - cb, d + sa - sb) (LRL): shared between the words by the caller.
RPP_HD static inline bool dubins_word(int wi, double al, double be, double d, const double* tg, double d2, double a1,
                                      double a2, double a3, double* o) {
  const double sa = tg[0], sb = tg[1], ca = tg[2], cb = tg[3], cab = tg[4];
  switch (wi) {
    case 0: {  // _LSL :1125-1135
      const double p2 = 2 + d2 - (2 * cab) + (2 * d * (sa - sb));
      if (p2 < 0) return false;
      const double tmp = a1;
      o[0] = mod2pi(-al + tmp); o[1] = __builtin_sqrt(p2); o[2] = mod2pi(be - tmp);
      return true;
    }
    case 1: {  // _RSR
      const double p2 = 2 + d2 - (2 * cab) + (2 * d * (sb - sa));
      if (p2 < 0) return false;
      const double tmp = a2;
      o[0] = mod2pi(al - tmp); o[1] = __builtin_sqrt(p2); o[2] = mod2pi(-be + tmp);
      return true;
    }
    case 2: {  // _LSR
      const double p2 = -2 + d2 + (2 * cab) + (2 * d * (sa + sb));
      if (p2 < 0) return false;
      const double d1 = __builtin_sqrt(p2);
      const double tmp = rpp_glibc_atan2((-ca - cb), (d + sa + sb)) - rpp_glibc_atan2(-2.0, d1);
      o[0] = mod2pi(-al + tmp); o[1] = d1; o[2] = mod2pi(-mod2pi(be) + tmp);
      return true;
    }
    case 3: {  // _RSL
      const double p2 = d2 - 2 + (2 * cab) - (2 * d * (sa + sb));
      if (p2 < 0) return false;
      const double d1 = __builtin_sqrt(p2);
      const double tmp = rpp_glibc_atan2((ca + cb), (d - sa - sb)) - rpp_glibc_atan2(2.0, d1);
      o[0] = mod2pi(al - tmp); o[1] = d1; o[2] = mod2pi(be - tmp);
      return true;
    }
    case 4: {  // _RLR
      const double tmp = (6.0 - d2 + 2.0 * cab + 2.0 * d * (sa - sb)) / 8.0;
      if (dabs(tmp) > 1.0) return false;
      const double dd2 = mod2pi(2 * kPi - rpp_glibc_acos(tmp));
      const double d1 = mod2pi(al - a2 + dd2 / 2.0);
      o[0] = d1; o[1] = dd2; o[2] = mod2pi(al - be - d1 + dd2);
      return true;
    }
    default: {  // _LRL
      const double tmp = (6.0 - d2 + 2.0 * cab + 2.0 * d * (-sa + sb)) / 8.0;
      if (dabs(tmp) > 1.0) return false;
      const double dd2 = mod2pi(2 * kPi - rpp_glibc_acos(tmp));
      const double d1 = mod2pi(-al - a3 + dd2 / 2.0);
      o[0] = d1; o[1] = dd2; o[2] = mod2pi(mod2pi(be) - al - d1 + mod2pi(dd2));
      return true;
    }
  }
}
// mode letter of segment s of word wi: 0 = L, 1 = S, 2 = R
RPP_HD static inline int dubins_mode(int wi, int s) {
  const int tab[6][3] = {{0, 1, 0}, {2, 1, 2}, {0, 1, 2}, {2, 1, 0}, {2, 0, 2}, {0, 2, 0}};
  return tab[wi][s];
}

// _interpolate :1232-1255 (local frame)
RPP_HD static inline void dubins_interp(double length, int mode, double maxc, double ox, double oy, double oyaw,
                                        double* x, double* y, double* yaw) {
  if (mode == 1) {
    *x = ox + length / maxc * rpp_glibc_cos(oyaw);
    *y = oy + length / maxc * rpp_glibc_sin(oyaw);
    *yaw = oyaw;
  } else {
    const double ldx = rpp_glibc_sin(length) / maxc;
    const double ldy = (mode == 0) ? (1.0 - rpp_glibc_cos(length)) / maxc : (1.0 - rpp_glibc_cos(length)) / -maxc;
    const double c = rpp_glibc_cos(-oyaw), s = rpp_glibc_sin(-oyaw);
    const double gdx = c * ldx + s * ldy;
    const double gdy = -s * ldx + c * ldy;
    *x = ox + gdx;
    *y = oy + gdy;
    *yaw = (mode == 0) ? oyaw + length : oyaw - length;
  }
}

// The terms of _interpolate that depend on the segment only: cos/sin of the origin yaw (straight) or of its negative
// (arcs), exactly as dubins_interp evaluates them.
RPP_HD static inline void dubins_seg_terms(int mode, double oyaw, double* cs, double* sn) {
  if (oyaw == 0.0) {   // cos(+-0) = 1, sin(+-0) = +-0 (the first segment always starts at yaw 0)
    *cs = 1.0;
    *sn = (mode == 1) ? oyaw : -oyaw;
  } else if (mode == 1) {
    *cs = rpp_glibc_cos(oyaw);
    *sn = rpp_glibc_sin(oyaw);
  } else {
    *cs = rpp_glibc_cos(-oyaw);
    *sn = rpp_glibc_sin(-oyaw);
  }
}
// _interpolate :1232-1255 with the segment terms and sin/cos of the length supplied (same operations, same order)
RPP_HD static inline void dubins_interp_t(double length, int mode, double maxc, double ox, double oy, double oyaw,
                                          double cs, double sn, double sl, double cl, double* x, double* y,
                                          double* yaw) {
  const bool unit = maxc == 1.0;   // v / 1.0 == v and v / -1.0 == -v exactly: skip the divisions
  if (mode == 1) {
    const double lm = unit ? length : length / maxc;
    *x = ox + lm * cs;
    *y = oy + lm * sn;
    *yaw = oyaw;
  } else {
    const double omc = 1.0 - cl;
    const double ldx = unit ? sl : sl / maxc;
    const double ldy = (mode == 0) ? (unit ? omc : omc / maxc) : (unit ? -omc : omc / -maxc);
    const double gdx = cs * ldx + sn * ldy;
    const double gdy = -sn * ldx + cs * ldy;
    *x = ox + gdx;
    *y = oy + gdy;
    *yaw = (mode == 0) ? oyaw + length : oyaw - length;
  }
}

// Everything of plan_dubins_path up to (not including) the per-point interpolation: local goal, word selection,
// per-segment origins, rotation terms, end points and point counts.  Points are then independent: point k of
// segment s is _interpolate(cur_k, mode_s, ...) with cur_k = step + step + ... (k additions, :1268-1273), or the
// segment end (which is also the next segment's origin, computed here).
struct DubinsPlan {
  double sx, sy, syaw;
  double rot_back[4];        // rot_mat_2d(-s_yaw)
  double len[3];             // b_d1..3 (in curvature units)
  double ox[3], oy[3], oyaw[3];   // local origin of each segment
  double cs[3], sn[3];            // dubins_seg_terms of each segment that has points
  double ex[3], ey[3], eyaw[3];   // local end of each segment (= _interpolate(len[s], ...))
  int32_t word, ok;
  int32_t mode[3];
  int32_t npts[3];           // points each segment contributes (0 when its length is 0)
  int32_t total;             // 1 (origin) + sum npts
};
constexpr double kDubinsStep = 0.1;

RPP_HD static inline void dubins_prepare(DubinsPlan* P, double sx, double sy, double syaw, double gx, double gy,
                                         double gyaw, double curv) {
  double lr[4], lrb[4];
  rot_mat_2d_pair(syaw, lr, lrb);
  const double p0 = gx - sx, p1 = gy - sy;
  const double lx = __builtin_fma(p1, lr[2], p0 * lr[0]);   // (2,) @ (2,2)  :1091-1093
  const double ly = __builtin_fma(p1, lr[3], p0 * lr[1]);
  const double lyaw = gyaw - syaw;
  const double d = py_hypot(lx, ly) * curv;                 // :1205
  const double theta = mod2pi(rpp_glibc_atan2(ly, lx));
  const double alpha = mod2pi(-theta), beta = mod2pi(lyaw - theta);
  double tg[5] = {rpp_glibc_sin(alpha), rpp_glibc_sin(beta), rpp_glibc_cos(alpha), rpp_glibc_cos(beta),
                  rpp_glibc_cos(alpha - beta)};
  const double d2 = py_sq(d);
  const double y1 = tg[3] - tg[2], y3 = tg[2] - tg[3];
  const double a1 = rpp_glibc_atan2(y1, d + tg[0] - tg[1]);
  const double a2 = rpp_glibc_atan2(y3, d - tg[0] + tg[1]);
  // atan2 is odd in y bit for bit; y3 == -y1 except when both are +0
  const double a3 = (d2b(y3) == d2b(-y1)) ? -a1 : rpp_glibc_atan2(y3, d + tg[0] - tg[1]);
  double best = dinf();
  int bw = -1;
  double bl0 = 0.0, bl1 = 0.0, bl2 = 0.0;
  for (int wi = 0; wi < 6; wi++) {
    double o[3];
    if (!dubins_word(wi, alpha, beta, d, tg, d2, a1, a2, a3, o)) continue;
    const double cost = dabs(o[0]) + dabs(o[1]) + dabs(o[2]);
    if (best > cost) {   // strict: the first word wins ties :1220
      best = cost;
      bw = wi;
      bl0 = o[0]; bl1 = o[1]; bl2 = o[2];
    }
  }
  P->len[0] = bl0; P->len[1] = bl1; P->len[2] = bl2;
  P->sx = sx; P->sy = sy; P->syaw = syaw;
  P->word = bw;
  P->ok = bw >= 0;
  P->total = 0;
  if (bw < 0) return;
  P->rot_back[0] = lrb[0]; P->rot_back[1] = lrb[1]; P->rot_back[2] = lrb[2]; P->rot_back[3] = lrb[3];
  double ox = 0.0, oy = 0.0, oyaw = 0.0;
  int total = 1;
  for (int s = 0; s < 3; s++) {
    P->ox[s] = ox; P->oy[s] = oy; P->oyaw[s] = oyaw;
    const double length = P->len[s];
    const int mode = dubins_mode(bw, s);
    P->mode[s] = mode;
    if (length == 0.0) {
      P->npts[s] = 0;
      P->cs[s] = P->sn[s] = 0.0;
      P->ex[s] = ox; P->ey[s] = oy; P->eyaw[s] = oyaw;
      continue;
    }
    int cnt = 0;
    double cur = kDubinsStep;
    while (dabs(cur + kDubinsStep) <= dabs(length)) {   // :1268-1273
      cnt++;
      cur += kDubinsStep;
    }
    P->npts[s] = cnt + 1;
    total += cnt + 1;
    double cs, sn;
    dubins_seg_terms(mode, oyaw, &cs, &sn);
    P->cs[s] = cs;
    P->sn[s] = sn;
    double sl = 0.0, cl = 0.0;
    if (mode != 1) {
      sl = rpp_glibc_sin(length);
      cl = rpp_glibc_cos(length);
    }
    dubins_interp_t(length, mode, curv, ox, oy, oyaw, cs, sn, sl, cl, &ox, &oy, &oyaw);   // segment end = next origin
    P->ex[s] = ox; P->ey[s] = oy; P->eyaw[s] = oyaw;
  }
  P->total = total;
}

// local-frame point j (0-based) of segment s; sl/cl = sin/cos of the interpolation length when the caller has them
// tabulated (arc, inner point), else computed here
RPP_HD static inline void dubins_seg_point(const DubinsPlan& P, int s, int j, double curv, double* lx, double* ly,
                                           double* lyaw) {
  if (j >= P.npts[s] - 1) {   // the segment end
    *lx = P.ex[s]; *ly = P.ey[s]; *lyaw = P.eyaw[s];
    return;
  }
  double cur = kDubinsStep;
  for (int q = 0; q < j; q++) cur += kDubinsStep;
  double sl = 0.0, cl = 0.0;
  if (P.mode[s] != 1) {
    sl = rpp_glibc_sin(cur);
    cl = rpp_glibc_cos(cur);
  }
  dubins_interp_t(cur, P.mode[s], curv, P.ox[s], P.oy[s], P.oyaw[s], P.cs[s], P.sn[s], sl, cl, lx, ly, lyaw);
}

// world-frame point number k (0 = start pose) of a prepared plan: (x, y, yaw) as plan_dubins_path returns them
RPP_HD static inline void dubins_point(const DubinsPlan& P, int k, double curv, double* wx, double* wy, double* wyaw) {
  double lx = 0.0, ly = 0.0, lyaw = 0.0;
  if (k > 0) {
    int s = 0, j = k - 1;
    while (j >= P.npts[s]) {
      j -= P.npts[s];
      s++;
    }
    dubins_seg_point(P, s, j, curv, &lx, &ly, &lyaw);
  }
  const double cx = __builtin_fma(ly, P.rot_back[2], lx * P.rot_back[0]);   // (n,2) @ (2,2)  :1103-1104
  const double cy = __builtin_fma(ly, P.rot_back[3], lx * P.rot_back[1]);
  *wx = cx + P.sx;
  *wy = cy + P.sy;
  *wyaw = angle_mod_pi(lyaw + P.syaw);   // :1107
}

}  // namespace rpp
