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
  int32_t uidx[NUI];
  int8_t ufree[NUI];        // -1 not tested, 0 blocked, 1 free
  int16_t urank[NUI];       // choose_parent: rank of a candidate by (cost, list position)
  int16_t tl[TPB];          // candidate slots whose segment is to be tested next (test_candidates)
  // uval (d**2 of a candidate, duplicate collapse) and cval (per-thread scratch of the same phase) are dead before
  // choose_parent writes ud / uex: they share storage, which keeps the block under 40 KB (4 workgroups per CU)
  union { double uval[NUI]; double ud[NUI]; };
  union { double cval[TPB]; double uex[NUI]; };
  double ux[NUI], uy[NUI], ucost[NUI], uey[NUI];
  int32_t cflag[TPB];
  double red_best[NW], red_second[NW];
  int32_t red_idx[NW], wave_cnt[NW], wave_start[NW];
  double rx, ry, nx, ny, ex, ey, ncost, cbest, plen;
  double rx2, ry2;          // sample of the NEXT iteration, drawn ahead without touching the RNG state (peek_sample)
  rpp::Sobol sob, sob2;     // Sobol state (any thread may draw), and the state after the peeked draw
  int32_t flag, nu, nvalid, overflow, ecoll, npar, nrw;
  int32_t pk_pos, pk_ok;    // MT19937 position after the peeked draw; 1 = the peek stayed inside the current 624-word block
  int32_t ntl, near_goal, nrw2;
  int16_t tlx[64];          // test_candidates: slots that need the exact (atan2 / cos / sin) form
  double ct, st;            // steer: cos(theta) from wave 0, sin(theta) from wave 2
};

// value of `v` in lane l of this wave (l wave-uniform); no LDS round trip: the all-pairs loops below walk a wave's registers
__device__ __forceinline__ double readlane_d(double v, int l) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
}

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
    double v = 0.0, hx = 0.0, hy = 0.0, hc = 0.0;
    bool valid = false;
    if (h < kraw) {
      idx = rppk::hit_at(hits, sh, h);
      hx = x[idx];
      hy = y[idx];
      hc = cost[idx];   // requested with the coordinates (one round trip), used only if the hit becomes a candidate
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
        sh.ucost[p] = hc;
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

// The same candidate records WITHOUT the two libm pow calls per hit, for the usual case in which no decision is close:
// vf = dx*dx + dy*dy (correctly rounded squares) is within 2^-51 relative of the reference's dx**2 + dy**2, so a hit whose
// vf is not within 2^-45 r**2 of r**2 has its `<= r**2` decided, and two hits whose vf differ by more than 2^-45 r**2 cannot
// hold the same reference value (the `.index` collapse of :1140-1142 needs equal values).  Returns false -- nothing usable
// written -- when some hit sits in the band or two hits are that close (exact duplicates included): the caller then runs
// build_candidates_i.  Otherwise the candidates are the valid hits in list order.
template <int NUI>
__device__ __forceinline__ bool build_candidates_fast(const double* __restrict__ x, const double* __restrict__ y,
                                                      const double* __restrict__ cost, double qx, double qy, double r2,
                                                      const int32_t* hits, int kraw, ShIT<NUI>& sh) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const double tol = r2 * 2.8421709430404007e-14;   // 2^-45 r**2
  if (tid == 0) {
    sh.nu = 0;
    sh.nvalid = 0;
    sh.nrw2 = 0;   // "a decision is close"
  }
  __syncthreads();
  for (int base = 0; base < kraw; base += TPB) {
    const int h = base + tid;
    int idx = -1;
    double v = rpp::b2d(0x7ff8000000000000ULL), hx = 0.0, hy = 0.0, hc = 0.0;
    bool valid = false, unsure = false;
    if (h < kraw) {
      idx = rppk::hit_at(hits, sh, h);
      hx = x[idx];
      hy = y[idx];
      hc = cost[idx];
      v = rpp::fast_d2(hx - qx, hy - qy);
      valid = v <= r2;
      unsure = rpp::dabs(v - r2) <= tol;
    }
    const int nu = sh.nu;
#pragma unroll 4
    for (int u = 0; u < nu; u++) unsure |= rpp::dabs(sh.uval[u] - v) <= tol;       // NaN (no hit): never close
    sh.cval[tid] = valid ? v : rpp::b2d(0x7ff8000000000000ULL);
    __syncthreads();
    const int nchunk = (kraw - base) < TPB ? (kraw - base) : TPB;
    {
      // every valid hit of the chunk against every other: with G = ceil(nchunk / 64) groups of 64 hits, wave w checks for
      // group w % G the partners t = part, part + P, ... (part = w / G of P = 4 / G parts), four LDS values per round trip
      const int nch = __builtin_amdgcn_readfirstlane(nchunk);   // wave-uniform by construction: scalar loop control
      const int G = (nch + 63) >> 6, P = G == 1 ? 4 : G == 2 ? 2 : 1;
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int g = wv % G, part = wv / G;
      if (part < P) {
        const int me = 64 * g + lane;
        const double mv_ = sh.cval[me];   // NaN unless a valid hit
        bool close = false;
        for (int t0 = 4 * part; t0 < nch; t0 += 4 * P) {
          const double c0 = sh.cval[t0], c1 = sh.cval[t0 + 1], c2 = sh.cval[t0 + 2], c3 = sh.cval[t0 + 3];   // NaN past the chunk
          close |= (t0 != me) & (rpp::dabs(c0 - mv_) <= tol);
          close |= (t0 + 1 != me) & (rpp::dabs(c1 - mv_) <= tol);
          close |= (t0 + 2 != me) & (rpp::dabs(c2 - mv_) <= tol);
          close |= (t0 + 3 != me) & (rpp::dabs(c3 - mv_) <= tol);
        }
        unsure |= close;
      }
    }
    const uint64_t mv = __ballot(valid);
    if (__ballot(unsure) != 0ull && lane == 0) sh.nrw2 = 1;
    if (lane == 0) sh.red_idx[w] = __popcll(mv);
    __syncthreads();
    if (sh.nrw2) return false;   // uniform: written before the barrier
    int off = nu;
#pragma unroll
    for (int k = 0; k < NW; k++)
      if (k < w) off += sh.red_idx[k];
    int tot = nu;
#pragma unroll
    for (int k = 0; k < NW; k++) tot += sh.red_idx[k];
    if (valid) {
      const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int p = off + __popcll(mv & lt_mask);
      if (p < NUI) {
        sh.uval[p] = v;
        sh.uidx[p] = idx;
        sh.ux[p] = hx;
        sh.uy[p] = hy;
        sh.ucost[p] = hc;
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (tot > NUI) {
        sh.overflow = 1;
        tot = NUI;
      }
      sh.nu = tot;
      sh.nvalid = tot;
    }
    __syncthreads();
  }
  return true;
}

// check_collision (rrt_07:1271-1276) of the candidate slots listed in sh.tl[0..nt) (nt <= TPB); result in sh.ufree[e]
// (1 free, 0 blocked).  The reference tests the segment from the candidate v to end = v + (cos theta, sin theta) * d with
// theta = atan2(new - v), d = hypot(new - v) (:1117-1119 / :1242-1244 arrive at the same (theta, d)): `end` is the new
// node up to the roundings of those four calls (a few ULP of the coordinates).  A wave takes a candidate, its lanes the
// obstacles, and evaluates the segment distance (:1249-1261) with the NEW NODE ITSELF as the end point:
//   * obstacles out of reach of the segment are dropped first: dist(p, segment) >= |p - midpoint| - half length;
//   * a distance**2 that clears `size**2` by more than `tol` on either side decides the comparison (:1267) for that obstacle
//     whatever the few-ULP difference of the end point and the roundings of the formula do (tol is ~1e4 times their bound);
//   * an obstacle inside the band sends the candidate to the exact form: atan2 / cos / sin replicas, cosine on a lane of
//     wave 0 and sine on a lane of wave 1, then the reference's expression -- a few times per 10^6 tests.
// So a verdict costs no libm-grade call in all but those cases (before: three calls deep per batch, lanes diverging).
template <int NUI>
__device__ __forceinline__ void test_candidates(const Ctx& c, ShIT<NUI>& sh, int nt, double nx, double ny, double tolmul) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) sh.nrw2 = 0;   // candidates that need the exact form
  __syncthreads();
  for (int t = w; t < nt; t += NW) {
    const int e = sh.tl[t];
    const double vx = sh.ux[e], vy = sh.uy[e], d = sh.ud[e];
    const double mx = 0.5 * (vx + nx), my = 0.5 * (vy + ny), hl = 0.5 * d * (1.0 + 1e-9) + 1e-9;
    const double sc = 1.0 + rpp::dabs(vx) + rpp::dabs(vy) + rpp::dabs(nx) + rpp::dabs(ny);
    bool hit = false, unsure = false;
    for (int k = lane; k < c.m; k += 64) {
      const double ddx = sh.ox[k] - mx, ddy = sh.oy[k] - my, tt = hl + sh.orad[k];
      if (ddx * ddx + ddy * ddy <= tt * tt * (1.0 + 1e-9)) {
        const double a = seg_dist2(vx, vy, nx, ny, sh.ox[k], sh.oy[k]);
        const double tol = 1e-10 * sc * (1.0 + d + sh.orad[k]) * tolmul;   // tolmul = 1; a test knob makes every verdict take the exact form
        if (a <= sh.othr[k] - tol)
          hit = true;
        else if (a <= sh.othr[k] + tol)
          unsure = true;
      }
    }
    const bool any = __ballot(hit) != 0ull, uns = __ballot(unsure) != 0ull;
    if (lane == 0) {
      sh.ufree[e] = any ? 0 : 1;
      if (!any && uns) sh.tlx[atomicAdd(&sh.nrw2, 1) & 63] = (int16_t)e;
    }
  }
  __syncthreads();
  const int nx2 = sh.nrw2;
  if (nx2 == 0) return;
  // exact form for the candidates left undecided (at most 64 per round trip: more cannot occur with nt <= 256 in practice,
  // and if they did the list wraps and the overwritten ones are found again by the caller's next test of an untested slot --
  // they keep ufree = 1 only if listed, so mark every undecided slot untested first)
  __syncthreads();
  if (nx2 > 64) {   // never observed; fall back to the exact form for the whole list
    for (int t0 = 0; t0 < nt; t0 += 64) {
      const int t = t0 + lane;
      if (w < 2 && t < nt) {
        const int e = sh.tl[t];
        const double th = rpp_glibc_atan2(ny - sh.uy[e], nx - sh.ux[e]);
        if (w == 0)
          sh.uex[e] = sh.ux[e] + rpp_glibc_cos(th) * sh.ud[e];
        else
          sh.uey[e] = sh.uy[e] + rpp_glibc_sin(th) * sh.ud[e];
      }
    }
    __syncthreads();
    for (int t = w; t < nt; t += NW) {
      const int e = sh.tl[t];
      const double vx = sh.ux[e], vy = sh.uy[e], ex2 = sh.uex[e], ey2 = sh.uey[e];
      bool hit = false;
      for (int k = lane; k < c.m; k += 64)
        if (seg_dist2(vx, vy, ex2, ey2, sh.ox[k], sh.oy[k]) <= sh.othr[k]) hit = true;
      const bool any = __ballot(hit) != 0ull;
      if (lane == 0) sh.ufree[e] = any ? 0 : 1;
    }
    __syncthreads();
    return;
  }
  if (w < 2 && lane < nx2) {
    const int e = sh.tlx[lane];
    const double th = rpp_glibc_atan2(ny - sh.uy[e], nx - sh.ux[e]);
    if (w == 0)
      sh.uex[e] = sh.ux[e] + rpp_glibc_cos(th) * sh.ud[e];
    else
      sh.uey[e] = sh.uy[e] + rpp_glibc_sin(th) * sh.ud[e];
  }
  __syncthreads();
  for (int t = w; t < nx2; t += NW) {
    const int e = sh.tlx[t];
    const double vx = sh.ux[e], vy = sh.uy[e], ex2 = sh.uex[e], ey2 = sh.uey[e];
    bool hit = false;
    for (int k = lane; k < c.m; k += 64)
      if (seg_dist2(vx, vy, ex2, ey2, sh.ox[k], sh.oy[k]) <= sh.othr[k]) hit = true;
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

// MT19937 read-ahead: the tempered words at mt[pos..] WITHOUT advancing or twisting the generator.  A draw that would run
// past the current 624-word block sets `bad` (the caller then draws the ordinary way at the start of the next iteration).
struct MTPeek {
  const uint32_t* mt;
  int32_t pos, bad;
};
__device__ __forceinline__ uint32_t mt_next(MTPeek* s) {
  if (s->pos >= 624) {
    s->bad = 1;
    return 0u;
  }
  uint32_t y = s->mt[s->pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= (y >> 18);
  return y;
}
__device__ __forceinline__ uint32_t mt_next(rpp::MT* s) { return rpp::mt_next(s); }
template <class R>
__device__ __forceinline__ double rnd01(R* s) {   // random.random(): rpp::mt_random over either generator view
  const uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

// informed_sample :1145-1159 (sample_unit_ball :1162-1171, sample_free_space(_sobol) :1173-1191) with best path length
// `cb`; consumes `rng` / `sob` in the reference's order.  R = rpp::MT (the real draw) or MTPeek (read-ahead).
template <class R>
__device__ __forceinline__ void informed_sample(R* rng, rpp::Sobol* sob, double cb, const InformedArgs& ia, const Ctx& c,
                                                double gx, double gy, double& rx, double& ry) {
  if (cb < rpp::dinf()) {
    const double r0 = cb / 2.0;
    const double r1 = __builtin_sqrt(rpp::py_sq(cb) - ia.c_min2) / 2.0;          // :1147
    double a = rnd01(rng), b = rnd01(rng);                                       // sample_unit_ball :1162-1171
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
  } else {
    uint32_t r = mt_next(rng) >> 25;                                             // random.randint(0, 100) :1173-1191
    while (r >= 101) r = mt_next(rng) >> 25;
    if ((int)r > c.goal_sample_rate) {
      if (c.sampler == 1) {
        double q[2];
        rpp::sobol_next(sob, q);
        rx = c.rand_min + q[0] * (c.rand_max - c.rand_min);
        ry = c.rand_min + q[1] * (c.rand_max - c.rand_min);
      } else {
        const double u0 = rnd01(rng);
        rx = c.rand_min + (c.rand_max - c.rand_min) * u0;
        const double u1 = rnd01(rng);
        ry = c.rand_min + (c.rand_max - c.rand_min) * u1;
      }
    } else {
      rx = gx;
      ry = gy;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Streaming pass over the 16-bit coordinate mirror xq[] (4 bytes per node: x16 | y16 << 16, Ctx::xq; the scheme of
// rrt_star_v2_body.inc scan2q, 256-thread shape): each wave streams a contiguous share with 16-byte non-temporal loads
// (lane -> 4 adjacent nodes), a ring of QDI loads in flight per lane; squared distances are exact integers in grid units
// (saturating 16-bit subtract + v_dot2_i32_i16).
//   NEAR:    indices with grid distance**2 <= thr about the packed point qq, appended in ascending order to hits[]
//            (wave w owns the slots starting at its range start; sh.wave_cnt / wave_start describe the segments);
//   NEAREST: smallest and second smallest grid distance**2 to the packed point sq and the node the smallest came from;
//            lowest index on ties.
// Both are SUPERSET / candidate answers: the callers decide on the f64 coordinates.
typedef uint32_t v4u_i __attribute__((ext_vector_type(4)));
typedef short s2v_i __attribute__((ext_vector_type(2)));
#ifndef RRTX_QDI
#define RRTX_QDI 4
#endif
#ifndef RRTX_QNT
#define RRTX_QNT 0
#endif
constexpr int QDI = RRTX_QDI;
constexpr int QSLOT_I = 256;
constexpr uint32_t QSAT_I = 32767u * 32767u;
constexpr int NEWNODE = -2;   // "group" of a nearest answer that is the node appended after the pass
__device__ __forceinline__ uint32_t qdist_i(uint32_t node, uint32_t query) {
  const s2v_i d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s2v_i, node), __builtin_bit_cast(s2v_i, query));
  return (uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false);
}
__device__ __forceinline__ uint32_t umin3_i(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t umed3_i(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }

template <bool NEAR, bool NEAREST, bool MASK>
__device__ __forceinline__ void scan_q16_slot(const v4u_i v, const int i0, const int n, const uint32_t qq, const uint32_t thr,
                                              const uint32_t sq, const uint64_t lt_mask, int32_t* __restrict__ hits,
                                              const int ws, int& cnt, uint32_t& best, uint32_t& second, int& bgrp) {
  if (NEAREST) {
    uint32_t d[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      d[j] = qdist_i(v[j], sq);
      if (MASK) d[j] = (i0 + j < n) ? d[j] : 0xffffffffu;
    }
    const uint32_t b0 = best;
    second = min(second, umed3_i(best, d[0], d[1]));
    best = umin3_i(best, d[0], d[1]);
    second = min(second, umed3_i(best, d[2], d[3]));
    best = umin3_i(best, d[2], d[3]);
    // the node itself (lowest of the group on ties), so that the caller needs no second look at the group
    const int j = d[0] == best ? 0 : d[1] == best ? 1 : d[2] == best ? 2 : 3;
    bgrp = best < b0 ? i0 + j : bgrp;
  }
  if (NEAR) {
    bool hh[4];
    uint64_t mm[4], any = 0ull;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      hh[j] = qdist_i(v[j], qq) <= thr && !(MASK && i0 + j >= n);
      mm[j] = __ballot(hh[j]);
      any |= mm[j];
    }
    if (any != 0ull) {
      int pos = cnt, tot = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        pos += __popcll(mm[j] & lt_mask);
        tot += __popcll(mm[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (hh[j]) hits[ws + pos++] = i0 + j;
      cnt += tot;
    }
  }
}

__device__ __forceinline__ v4u_i qload(const v4u_i* p) {
#if RRTX_QNT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <bool NEAR, bool NEAREST, class SH>
__device__ __forceinline__ int scan_q16(const uint32_t* __restrict__ xq, int n, uint32_t qq, uint32_t thr, uint32_t sq,
                                        int32_t* __restrict__ hits, SH& sh, int& ggrp, uint32_t& gbest, uint32_t& gsecond) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int per = rppk::roundup_i((n + NW - 1) / NW, QSLOT_I);
  const int ws = w * per;
  const int wend = (ws + per < n) ? ws + per : n;
  const int nsl = wend > ws ? (wend - ws + QSLOT_I - 1) / QSLOT_I : 0;   // slots of this wave
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int cnt = 0;
  uint32_t best = 0xffffffffu, second = 0xffffffffu;
  int bgrp = 0x7ffffffc;
  if (nsl > 0) {
    const v4u_i* pv = reinterpret_cast<const v4u_i*>(xq + ws) + lane;   // slot s = pv[64 * s]
    const int last = nsl - 1;
    const int rounds = (nsl + QDI - 1) / QDI;
    v4u_i q[QDI];
#pragma unroll
    for (int u = 0; u < QDI; u++) q[u] = qload(pv + 64 * (u < last ? u : last));
    int s0 = 0;
    for (int r = 0; r + 1 < rounds; r++, s0 += QDI) {
#pragma unroll
      for (int u = 0; u < QDI; u++) {
        const int sl = s0 + u;
        scan_q16_slot<NEAR, NEAREST, false>(q[u], ws + sl * QSLOT_I + lane * 4, n, qq, thr, sq, lt_mask, hits, ws, cnt, best,
                                            second, bgrp);
        const int nx = sl + QDI;
        q[u] = qload(pv + 64 * (nx < last ? nx : last));   // unconditional, clamped to the last slot
      }
    }
#pragma unroll
    for (int u = 0; u < QDI; u++) {
      const int sl = s0 + u;
      if (sl < nsl)
        scan_q16_slot<NEAR, NEAREST, true>(q[u], ws + sl * QSLOT_I + lane * 4, n, qq, thr, sq, lt_mask, hits, ws, cnt, best,
                                           second, bgrp);
    }
  }
  if (NEAR && lane == 0) {
    sh.wave_cnt[w] = cnt;
    sh.wave_start[w] = ws;
  }
  if (NEAREST) {
    double gb, gs;
    rppk::block_argmin((double)best, bgrp, (double)second, sh, gb, ggrp, gs);   // contains the barriers that publish wave_cnt
    gbest = (uint32_t)gb;
    gsecond = (uint32_t)gs;
  } else {
    __syncthreads();
  }
  int total = 0;
  if (NEAR) {
#pragma unroll
    for (int k = 0; k < NW; k++) total += sh.wave_cnt[k];
  }
  return total;
}

template <int NUI, int WPS>
__global__ __launch_bounds__(TPB, WPS) void rrt_informed_kernel(Ctx c, const InformedArgs* __restrict__ per_inst,
                                                                double* cbest_io, int iters, int eager) {
  __shared__ ShIT<NUI> sh;
  const int inst = c.inst_map ? c.inst_map[blockIdx.x] : blockIdx.x;
  const InformedArgs ia = per_inst[inst];   // rotation, centre and c_min**2 of THIS instance's start / goal pair
  const int tid = threadIdx.x;
  Inst* I = c.inst + inst;
  if (I->status & 1) return;
  // test knobs (RRTX_INFORMED_EXACT_SEG=1): bit 1 = every near obstacle goes through the exact segment form, bit 2 = every
  // candidate list through the exact `**2` form
  const double tolmul = (eager & 2) ? 1e30 : 1.0;
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
  // 16-bit mirror (first stage of both queries, ONE pass per iteration): valid while every node lies inside the grid
  // square [q_lo, q_hi]^2 (sampling square + margin, rrtx_api.hip); informed samples and the nodes that follow them may
  // leave it, the instance then continues on the f32 / f64 passes for good (Inst::goal_dups = 1 carries that across launches)
  uint32_t* __restrict__ xq = c.xq ? c.xq + off : nullptr;
  int q16_ok = (c.xq != nullptr) && (I->goal_dups == 0);
  const double q_glo = c.q_lo + c.q_step, q_ghi = c.q_lo + 65534.0 * c.q_step;
  const double QMARGIN = 2.875 + 1e-6;   // 2 q_m in grid steps (q_m = 1.4375 steps) + slack for the two sqrt roundings
  if (!(I->start[0] >= q_glo && I->start[0] <= q_ghi && I->start[1] >= q_glo && I->start[1] <= q_ghi)) q16_ok = 0;

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
    sh.sob = I->sobol;
  }
  __syncthreads();
  int n = I->n, it = I->it;
  const double gx = I->goal[0], gy = I->goal[1], sx0 = I->start[0], sy0 = I->start[1];
  const double E = c.expand_dis;
  // Read-ahead across iterations (never across launches): have_s = this iteration's sample was drawn during the previous
  // one, have_n = its nearest query was answered by the previous iteration's pass: (p_best, p_second) grid distances,
  // p_grp the node that holds the best (NEWNODE: the node appended after that pass)
  int have_s = 0, have_n = 0, p_grp = 0;
  uint32_t p_best = 0u, p_second = 0u;
  int64_t s_qfb = 0;
  int64_t s_iter = 0, s_eu = 0, s_er = 0, s_nh = 0, s_nu = 0, s_rw = 0, s_sn = 0, s_ab = 0, s_ab2 = 0, s_ex = 0;
  int stop = 0;
  PH_DECL

  for (int step = 0; step < iters && it < c.max_iter && !stop; step++, it++) {
    s_iter++;
    PH(15);
    // ---------------- informed_sample :1145-1159 (unless the previous iteration drew it ahead: sh.rx2 / ry2, committed there)
    if (!have_s) {
      if (tid == 0) {
        double rx, ry;
        informed_sample(&sh.rng, &sh.sob, sh.cbest, ia, c, gx, gy, rx, ry);
        sh.rx = rx;
        sh.ry = ry;
      }
      __syncthreads();
    }
    const double rx = sh.rx, ry = sh.ry;

    PH(0);
    // ---------------- nearest :1210-1214
    int ni = -1;
    double gbest = 1.0, gsecond = rpp::dinf();
    bool nearest_done = false;
    {
      // first stage on the 16-bit mirror: the prefetched answer, or a pass of its own
      bool have_q = false;
      uint32_t qb = 0u, qs = 0u, sq = 0u;
      int qg = 0;
      const bool s_in = rx >= q_glo && rx <= q_ghi && ry >= q_glo && ry <= q_ghi;
      if (q16_ok && s_in) {
        sq = rppk::quant16(c, rx, ry);
        if (have_n) {
          qb = p_best;
          qs = p_second;
          qg = p_grp;
        } else {
          scan_q16<false, true>(xq, n, 0u, 0u, sq, hits, sh, qg, qb, qs);
          s_ab += 4 * (int64_t)n;
        }
        have_q = true;
      }
      if (have_q && qb < QSAT_I) {
        // node and query each sit within half a grid step per coordinate of their true position: a true distance differs
        // from the grid distance by less than q_m = 1.4375 steps, so the grid argmin is the true one when the runner-up is
        // more than 2 q_m further; otherwise every node within 2 q_m of the best grid distance is a candidate and the
        // decision is taken on the f64 coordinates with the reference's own expression (first minimum, :1212-1213)
        const double rb = __builtin_sqrt((double)qb);
        if (__builtin_sqrt((double)qs) - rb > QMARGIN) {
          ni = (qg == NEWNODE) ? n - 1 : qg;
          nearest_done = ni >= 0 && ni < n;
        }
        if (!nearest_done) {
          s_qfb++;
          const double rr = rb + QMARGIN;
          const double t2 = rr * rr * (1.0 + 1e-9) + 1.0;
          const uint32_t thr2 = t2 >= 4294967295.0 ? 0xffffffffu : (uint32_t)t2;
          int g0;
          uint32_t b0, b1;
          const int kraw = scan_q16<true, false>(xq, n, sq, thr2, 0u, hits, sh, g0, b0, b1);
          s_ab += 4 * (int64_t)n + 16 * (int64_t)kraw;
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
          nearest_done = ni >= 0 && ni < n;
        }
      }
    }
    if (!nearest_done) {
      const bool q_ok = f32_ok && rpp::dabs(rx) <= fmax && rpp::dabs(ry) <= fmax;
      bool f32_done = false;
      if (q_ok) {
        int fi;
        double fb, fs;
        rppk::scan_nearest_f32(xf, yf, n, (float)rx, (float)ry, sh, fi, fb, fs);
        s_ab += 8 * (int64_t)n;
        if (__builtin_sqrt(fs) - __builtin_sqrt(fb) > 2.0 * fm) {
          ni = fi;
          gbest = 1.0;
          gsecond = rpp::dinf();
          f32_done = true;
        }
      }
      if (!f32_done) {
        rppk::scan_nearest(x, y, n, rx, ry, sh, ni, gbest, gsecond);
        s_ab += 16 * (int64_t)n;
      }
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
    }
    s_sn += n;
    s_ab += 24 * (int64_t)c.m;
    s_ab2 += 16 * (int64_t)n + 24 * (int64_t)c.m;
    have_s = 0;
    have_n = 0;

    PH(1);
    // ---------------- steer :1080-1083 / get_new_node :1216-1224, extension edge :1085
    // lane 0: atan2 -> cos; lane 0 of wave 2: the same atan2 -> sin (the replicas are pure functions: both lanes hold the
    // same theta); lane 0 of wave 1: the read-ahead draw.  Then lane 0 finishes the node, lane 0 of wave 3 the goal test.
    const double r2n = c.r2tab[n];   // near radius of this iteration (:1139), requested here, used after the steer
    const double qx = x[ni], qy = y[ni];
    if (tid == 0) {
      const double theta = rpp_glibc_atan2(ry - qy, rx - qx);
      sh.ct = rpp_glibc_cos(theta);
    } else if (tid == 128) {
      const double theta = rpp_glibc_atan2(ry - qy, rx - qx);
      sh.st = rpp_glibc_sin(theta);
    } else if (tid == 64 && step + 1 < iters && it + 1 < c.max_iter) {
      // the sample of iteration it+1, read ahead on another wave while lane 0 steers: it depends on the RNG / Sobol state
      // and on c_best only.  Nothing is consumed here; the draw is committed at the end of this iteration if c_best is
      // still the value used (it changes only when this iteration's node connects to the goal with a shorter path)
      MTPeek pk{sh.rng.mt, sh.rng.pos, 0};
      rpp::Sobol sb = sh.sob;
      double ax, ay;
      informed_sample(&pk, &sb, sh.cbest, ia, c, gx, gy, ax, ay);
      sh.rx2 = ax;
      sh.ry2 = ay;
      sh.sob2 = sb;
      sh.pk_pos = pk.pos;
      sh.pk_ok = pk.bad ? 0 : 1;
      sh.plen = sh.cbest;   // c_best the read-ahead used
    }
    if (tid == 64 && !(step + 1 < iters && it + 1 < c.max_iter)) sh.pk_ok = 0;
    __syncthreads();
    PH(10);
    {
      const double ct = sh.ct, st = sh.st;
      const double nx0 = qx + E * ct, ny0 = qy + E * st;         // get_new_node :1219-1220
      if (tid == 0) {
        const double d = rpp::py_hypot(qx - nx0, qy - ny0);      // line_cost :1206
        sh.nx = nx0;
        sh.ny = ny0;
        sh.ux[0] = qx;                                            // segment start (candidate slot 0 is free here)
        sh.uy[0] = qy;
        sh.ex = qx + ct * d;                                      // check_collision :1273-1274
        sh.ey = qy + st * d;
        sh.ncost = cost[ni] + E;                                  // :1222
        sh.npar = ni;
        sh.ecoll = 0;
      } else if (tid == 192) {
        sh.near_goal = (rpp::py_hypot(nx0 - gx, ny0 - gy) < E) ? 1 : 0;   // is_near_goal :1226-1230 (strict), used by :1094
      }
    }
    __syncthreads();
    PH(11);
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
      const double r2 = r2n;
      int kraw;
      const bool n_in = nx >= q_glo && nx <= q_ghi && ny >= q_glo && ny <= q_ghi;
      if (q16_ok && !n_in) q16_ok = 0;   // the new node cannot be represented on the grid: f32 / f64 passes from here on
      if (q16_ok) {
        // ONE pass answers the near-ball query of this iteration and -- when the next sample is known (read-ahead valid)
        // and lies on the grid -- the nearest query of the next: superset ball {grid distance <= r + q_m}
        const double rg = (__builtin_sqrt(r2) + c.q_m) * c.q_inv;
        const double t2 = rg * rg * (1.0 + 1e-9) + 1.0;
        const uint32_t thr = t2 >= 4294967295.0 ? 0xffffffffu : (uint32_t)t2;
        const uint32_t qq = rppk::quant16(c, nx, ny);
        PH(12);
        const double ax = sh.rx2, ay = sh.ry2;
        const bool pre = sh.pk_ok && ax >= q_glo && ax <= q_ghi && ay >= q_glo && ay <= q_ghi;
        if (pre) {
          kraw = scan_q16<true, true>(xq, n, qq, thr, rppk::quant16(c, ax, ay), hits, sh, p_grp, p_best, p_second);
          have_n = 1;
        } else {
          int g0;
          uint32_t b0, b1;
          kraw = scan_q16<true, false>(xq, n, qq, thr, 0u, hits, sh, g0, b0, b1);
        }
        s_ab += 4 * (int64_t)n + 16 * (int64_t)kraw;
      } else if (f32_ok && rpp::dabs(nx) <= fmax && rpp::dabs(ny) <= fmax) {
        const double rr = __builtin_sqrt(r2) + fm;   // ball radius r + m in the f32 metric, rounded up
        kraw = rppk::scan_hits_f32(xf, yf, n, (float)nx, (float)ny, (float)(rr * rr * (1.0 + 1e-6)), hits, sh);
        s_ab += 8 * (int64_t)n + 16 * (int64_t)kraw;
      } else {
        kraw = rppk::scan_hits(x, y, n, nx, ny, r2 * (1.0 + FILTER_EPS), hits, sh);
        s_ab += 16 * (int64_t)n;
      }
      s_sn += n;
      PH(3);
      if ((eager & 4) || !build_candidates_fast(x, y, cost, nx, ny, r2, hits, kraw, sh)) {
        __syncthreads();
        build_candidates_i(x, y, cost, nx, ny, r2, hits, kraw, sh);
        s_ex++;
      }
      const int nu = sh.nu, nvalid = sh.nvalid;
      nnear = nu;
      s_nh += nvalid;
      s_nu += nu;
      s_ab += 48 * (int64_t)nu + 28;
      s_ab2 += 16 * (int64_t)n + 48 * (int64_t)nu + 28;
      PH(4);
      if (!(eager & 1)) {
        // ---------------- choose_parent :1110-1135, cheapest first.  A candidate's cost `cost_i + d` (:1120) does not
        // depend on its collision test, so the candidates are ranked by (cost, list position) and tested in growing
        // batches (4, 8, 16 ...): the first free one in rank order IS the reference's `min(d_list)` / `.index` pick
        // (:1125-1126); later ones are never tested.  atan2 / cos / sin (:1118, :1273-1274) are evaluated for tested
        // candidates only.  ufree: -1 not tested, 0 blocked, 1 free.  The rank lives in uey[] until the slot is tested.
        int32_t* racc = reinterpret_cast<int32_t*>(sh.uey);   // rank accumulators (uey is free until a segment needs its exact form)
        for (int e = tid; e < nu; e += TPB) {
          const double de = rpp::py_hypot(nx - sh.ux[e], ny - sh.uy[e]);
          sh.ud[e] = de;
          sh.uex[e] = sh.ucost[e] + de;   // the candidate's cost :1120 (uex is free until a segment needs its exact form)
          sh.ufree[e] = -1;
          racc[e] = 0;
        }
        if (tid < 8 && nu + tid < NUI) sh.uex[nu + tid] = rpp::dinf();   // the rank loops read whole blocks of 4
        if (tid == 0) {
          sh.ntl = 0;
          sh.flag = 0x7fffffff;
        }
        __syncthreads();
        PH(7);
        // rank + the first batch: the 4 cheapest candidates, and with them the candidates rewire (:1232-1246) will most
        // likely ask for -- those whose cost would improve if the cheapest candidate becomes the parent (the usual case).
        // That guess only decides WHEN a segment is tested: rewire below still tests whatever it needs and has no verdict for
        constexpr int B0 = 4;
        {
          const int nus = __builtin_amdgcn_readfirstlane(nu);   // wave-uniform by construction: scalar loop control
          if (nus <= TPB) {
            // rank of candidate e = number of (cost, position) pairs below its own.  All four waves share the pairs: with
            // G = ceil(nu / 64) groups of 64 candidates, wave w counts for group w % G the competitors j = part, part + P, ...
            // (part = w / G of P = 4 / G parts), four LDS values per round trip, and adds its count to the candidate's
            // accumulator; ties (equal costs) are rare and settled by position in the same loop
            const int G = (nus + 63) >> 6, P = G == 1 ? 4 : G == 2 ? 2 : 1;
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int g = wv % G, part = wv / G;
            double cmin = rpp::dinf();
            if (part < P) {
              const int e = 64 * g + (tid & 63);
              const double ce = e < nus ? sh.uex[e] : rpp::dinf();
              int rk = 0;
              for (int j0 = 4 * part; j0 < nus; j0 += 4 * P) {
                const double c0 = sh.uex[j0], c1 = sh.uex[j0 + 1], c2 = sh.uex[j0 + 2], c3 = sh.uex[j0 + 3];   // padded with +inf
                rk += (int)(c0 < ce) | ((int)(c0 == ce) & (int)(j0 < e));
                rk += (int)(c1 < ce) | ((int)(c1 == ce) & (int)(j0 + 1 < e));
                rk += (int)(c2 < ce) | ((int)(c2 == ce) & (int)(j0 + 2 < e));
                rk += (int)(c3 < ce) | ((int)(c3 == ce) & (int)(j0 + 3 < e));
                const double m01 = c0 < c1 ? c0 : c1, m23 = c2 < c3 ? c2 : c3, m = m01 < m23 ? m01 : m23;
                cmin = m < cmin ? m : cmin;
              }
              if (e < nus && rk) atomicAdd(&racc[e], rk);
            }
            if ((tid & 63) == 0) sh.red_best[wv] = cmin;   // this wave's share of the cheapest cost
            __syncthreads();
            if (tid < nus) {
              const int e = tid, rk = racc[e];
              double cm = sh.red_best[0];
#pragma unroll
              for (int k = 1; k < NW; k++) cm = sh.red_best[k] < cm ? sh.red_best[k] : cm;
              sh.urank[e] = (int16_t)rk;
              if (rk < B0) sh.cflag[rk] = e;
              if (rk < B0 || sh.ucost[e] > cm + sh.ud[e]) sh.tl[atomicAdd(&sh.ntl, 1)] = (int16_t)e;   // <= nu entries
            }
          } else {
            for (int e = tid; e < nu; e += TPB) {
              const double ce = sh.uex[e];
              int rk = 0;
#pragma unroll 4
              for (int j = 0; j < nus; j++) {
                const double cj = sh.uex[j];
                rk += (int)(cj < ce) | ((int)(cj == ce) & (int)(j < e));
              }
              sh.urank[e] = (int16_t)rk;
              if (rk < B0) {
                sh.cflag[rk] = e;
                sh.tl[atomicAdd(&sh.ntl, 1)] = (int16_t)e;
              }
            }
          }
        }
        __syncthreads();
        PH(5);
        int tested = 0, found = 0;
        for (int base = 0, bsz = B0; base < nu && !found; base += bsz, bsz = bsz < TPB / 2 ? 2 * bsz : TPB) {
          const int na = (nu - base < bsz) ? nu - base : bsz;
          if (base > 0) {
            // a later batch (the whole first one was blocked): ranks base .. base + na - 1, minus what has a verdict already
            if (tid == 0) {
              sh.ntl = 0;
              sh.flag = 0x7fffffff;
            }
            __syncthreads();
            for (int e = tid; e < nu; e += TPB) {
              const int rk = sh.urank[e];
              if (rk >= base && rk < base + na) {
                sh.cflag[rk - base] = e;
                if (sh.ufree[e] < 0) sh.tl[atomicAdd(&sh.ntl, 1)] = (int16_t)e;
              }
            }
            __syncthreads();
          }
          const int nt = sh.ntl;
          test_candidates(c, sh, nt, nx, ny, tolmul);
          tested += nt;
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
        // the candidates are independent.  Those without a verdict yet are tested now.
        if (tid == 0) {
          x[n] = nx;
          y[n] = ny;
          if (xf) {
            xf[n] = (float)nx;
            yf[n] = (float)ny;
          }
          if (xq) xq[n] = rppk::quant16(c, nx, ny);
          cost[n] = ncost;
          parent[n] = sh.npar;
          sh.nrw = 0;
          sh.ntl = 0;
        }
        __syncthreads();
        int extra = 0;
        for (int e0 = 0; e0 < nu; e0 += TPB) {
          const int e = e0 + tid;
          const bool need = e < nu && sh.ucost[e] > ncost + sh.ud[e] && sh.ufree[e] < 0;
          if (need) sh.tl[atomicAdd(&sh.ntl, 1)] = (int16_t)e;
          __syncthreads();
          const int nt = sh.ntl;
          __syncthreads();
          if (nt > 0) {
            test_candidates(c, sh, nt, nx, ny, tolmul);
            extra += nt;
            if (tid == 0) sh.ntl = 0;
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
        if (xq) xq[n] = rppk::quant16(c, nx, ny);
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
      if (sh.near_goal) {   // is_near_goal :1226-1230, evaluated beside the steer (sh.ecoll is 0 here: the node was accepted)
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
    // ---------------- commit the read-ahead: valid when the draw stayed inside the MT block and c_best is still the value it
    // used (bitwise: c_best only ever changes to a smaller finite value)
    const bool keep = sh.pk_ok && rpp::d2b(sh.plen) == rpp::d2b(sh.cbest) && !stop;
    if (keep && have_n && accepted) {
      // the node appended after the pass takes part in the prefetched nearest answer with its own grid distance
      const uint32_t dn = qdist_i(rppk::quant16(c, nx, ny), rppk::quant16(c, sh.rx2, sh.ry2));
      if (dn < p_best) {
        p_second = p_best;
        p_best = dn;
        p_grp = NEWNODE;
      } else if (dn < p_second) {
        p_second = dn;
      }
    }
    __syncthreads();
    if (keep) {
      if (tid == 0) {
        sh.rx = sh.rx2;
        sh.ry = sh.ry2;
        sh.rng.pos = sh.pk_pos;
        sh.sob = sh.sob2;
      }
      have_s = 1;
    } else {
      have_n = 0;
    }
    __syncthreads();
  }

  __syncthreads();
  for (int i = tid; i < 624; i += TPB) I->rng.mt[i] = sh.rng.mt[i];
  if (tid == 0) {
    I->rng.pos = sh.rng.pos;
    I->sobol = sh.sob;
    I->n = n;
    I->it = it;
    if (!f32_ok) I->first_goal = -3;
    if (!q16_ok && c.xq) I->goal_dups = 1;
    I->q16_fallbacks += s_qfb;
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
