// rrt_kernels.hip.h -- gfx950 kernels of the batched RRT / RRT* planner.
//
// One 256-thread workgroup (4 wave64) owns one planning instance and runs its
// iterations back to back on the device; many instances are resident per GPU
// (4 workgroups per CU), which is what turns the two O(n) node-array scans of
// every iteration into an HBM-streaming workload.
//
// HBM layout per instance (SoA, stride `stride` nodes, padded with +inf so the
// scans need no bounds checks):  x[], y[] f64 (the only arrays the scans read:
// 16 B/node), cost[] f64, parent[] i32, first_child/next_sib/prev_sib i32 (child
// lists = the identity scans of rrt_04:1369-1371 and :1381-1384 with integer
// parents), hits[] / stack[] i32 scratch.
//
// Per-iteration phases (reference: 10_path_planning_01_rrt_04_rrt_star.py)
//   sample        lane 0, MT19937 / Sobol                     :1132-1153
//   nearest       all lanes, wave-contiguous streaming argmin  :1197-1202
//   steer+collide lane 0 steer, lanes over obstacles (LDS)     :1086-1115, :1216-1230
//   near          all lanes, streaming threshold scan with ballot-ordered
//                 compaction, exact **2 re-check, `.index` de-dup   :1314-1338
//   choose_parent lanes over candidates x obstacles            :1242-1282
//   rewire        parallel steer/collision, sequential cost resolution and
//                 cost propagation through child lists         :1340-1384
//   goal search   streaming scan + candidates                  :1284-1312
//
// Exactness: the scans filter with correctly rounded dx*dx+dy*dy, which is within
// 2^-51 (relative) of the reference's dx**2+dy**2 (glibc pow, < 1 ULP); every
// decision inside the filter margin 2^-47 is re-taken with the exact replica
// (rpp::py_d2 / rpp::py_hypot), so integer results equal the reference's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rpp_core.h"
#include "rpp_rs.h"

namespace rppk {

constexpr int TPB = 256;
constexpr int NW = TPB / 64;
constexpr int UNROLL = 4;
constexpr int WAVE_STRIDE = 128 * UNROLL;  // nodes one wave consumes per loop trip
constexpr int MAX_OBS = 256;
constexpr int NU_MAX = 512;   // distinct near candidates kept in LDS
constexpr int EB = 64;        // edges steered per pass (single-direction form)
constexpr int EBD = 32;       // candidates per pass of the two-direction form (2*EBD edge slots)
constexpr int FCAP = TPB;     // BFS frontier capacity in LDS (2 buffers inside Sh::cval)
constexpr double FILTER_EPS = 7.105427357601002e-15;  // 2^-47

struct Result {  // 16 B record gathered across GPUs
  double path_cost;
  int32_t n_nodes;
  int32_t status;
};

struct Inst {  // persistent per-instance state (global memory)
  rpp::MT rng;
  rpp::Sobol sobol;
  double start[3], goal[3];   // x, y and (pose planners: rrt_03 / rrt_05 / rrt_06) yaw of this instance
  int32_t n, it, status, goal_node, path_n;
  int32_t first_goal;   // lowest index of a node lying exactly on the goal (-1 none yet, -2 unknown); f32-mirror path only
  int32_t goal_dups;    // nodes with a higher index lying exactly on the goal too, while first_goal >= 0
  int64_t iterations, edges_unique, edges_ref, near_hits, near_unique, rewires, propagated, scan_nodes, alg_bytes,
      exact_rescans, alg_bytes2, nu_max, f32_fallbacks, q16_fallbacks, rides;
  int64_t phase[16];  // shader-clock cycles per phase as lane 0 sees them (filled by -DRRTX_PHASE_TIMERS builds only)
};

#ifdef RRTX_PHASE_TIMERS
#define PH_DECL int64_t ph_[16] = {0}; int64_t pt0_ = (int64_t)__builtin_amdgcn_s_memtime();
#define PH(k) do { if (threadIdx.x == 0) { int64_t t_ = (int64_t)__builtin_amdgcn_s_memtime(); ph_[k] += t_ - pt0_; pt0_ = t_; } } while (0)
#define PH_STORE(I) do { for (int k_ = 0; k_ < 16; k_++) (I)->phase[k_] += ph_[k_]; } while (0)
#else
#define PH_DECL
#define PH(k) do { } while (0)
#define PH_STORE(I) do { } while (0)
#endif

struct Ctx {
  Inst* inst;
  double *x, *y, *cost;
  int32_t *parent, *first_child, *next_sib, *prev_sib, *hits, *stack;
  int64_t stride;
  const double *ox, *oy, *othr;
  int32_t m;
  const double* r2tab;
  double* path_xy;
  int32_t path_cap;
  Result* results;
  int32_t algo, sampler, goal_sample_rate, max_iter, has_play, until_max;
  double rand_min, rand_max, expand_dis, res;
  double play_area[4];
  int32_t trace_inst;
  double *tr_rx, *tr_ry;
  int32_t *tr_near, *tr_nn;
  int32_t* tr_kind;   // per iteration: 0 no node appended, 1 the extension itself (rrt_01:85-96, rrt_04:1066-1067), 2 under a chosen parent (:1062-1065)
  // f32 mirror of x[], y[] (prefilter of the streaming pass, rrt_star_v2_body.inc) and its distance margin
  float *xf, *yf;
  double f32_m;
  // 16-bit fixed-point mirror, 4 bytes per node: (x16 | y16 << 16), q = rint((coord - q_lo) * q_inv) - 32768; first stage of
  // the rrt_04 iteration kernel's streaming pass (scan2q); q_m = distance margin of that stage
  uint32_t* xq;
  double q_lo, q_inv, q_step, q_m;
  // partial re-plan (overflow retry, rrtx_api.hip): block b works on instance inst_map[b]; nullptr = identity
  const int32_t* inst_map;
  // elen[i] = hypot(node i - its parent), exactly the value calc_new_cost (rrt_04:1375-1377) would compute now;
  // kept by the v2 kernel so cost propagation needs no coordinates and no hypot
  double* elen;
  // two iterations per streaming pass in the one-wave shape of the rrt_04 iteration kernel (rrt_star_v2_body.inc)
  int32_t spec2;
};

// 16-bit mirror entry of a point (Ctx::xq)
__device__ __forceinline__ uint32_t quant16(const Ctx& c, double px, double py) {
  double qx = __builtin_rint((px - c.q_lo) * c.q_inv), qy = __builtin_rint((py - c.q_lo) * c.q_inv);
  qx = qx < 0.0 ? 0.0 : (qx > 65535.0 ? 65535.0 : qx);
  qy = qy < 0.0 ? 0.0 : (qy > 65535.0 ? 65535.0 : qy);
  return ((uint32_t)qx | ((uint32_t)qy << 16)) ^ 0x80008000u;   // each half as a signed 16-bit value: q - 32768
}

struct Sh {
  static constexpr int kNU = NU_MAX, kTPB = TPB, kNW = NW;   // the templated helpers below take their shape from the LDS struct
  rpp::MT rng;
  double ox[MAX_OBS], oy[MAX_OBS], othr[MAX_OBS];
  rpp::Edge edge[EB];
  int32_t ecoll[EB];
  double uval[NU_MAX];
  int32_t uidx[NU_MAX];
  double uex[NU_MAX], uey[NU_MAX], uaux[NU_MAX];
  int32_t usafe[NU_MAX];
  double cval[TPB];
  int32_t cflag[TPB];
  double red_best[NW], red_second[NW];
  int32_t red_idx[NW];
  int32_t wave_cnt[NW];
  int32_t wave_start[NW];
  double rx, ry, nx, ny, ncost, wx, wy, wcost;
  int32_t ni, flag, nu, nvalid, sel, overflow;
  int32_t fa, fb, fover, fcount;   // BFS frontier sizes / overflow / visited count (propagate_bfs)
};

__device__ __forceinline__ int roundup_i(int a, int b) { return (a + b - 1) / b * b; }

// 16-byte streaming load of two adjacent nodes.  The node arrays are re-read from HBM every iteration (the
// per-GPU working set is far larger than L2 / Infinity Cache), so they are loaded non-temporally to leave the
// caches to the small hot data: libm tables, child lists, candidate gathers.
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2d stream2(const double* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));
}

// ---------------------------------------------------------------------------
// block-wide (value, index) argmin with lowest-index tie break; `second` is the
// smallest value held by any element other than the winner (for the filter).
template <class SH>
__device__ __forceinline__ void block_argmin(double best, int bidx, double second, SH& sh, double& gbest, int& gidx,
                                             double& gsecond) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double b = best;
  int bi = bidx;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    double ob = __shfl_xor(b, o);
    int oi = __shfl_xor(bi, o);
    bool take = (ob < b) || (ob == b && oi < bi);
    b = take ? ob : b;
    bi = take ? oi : bi;
  }
  double s = (bidx == bi) ? second : best;  // best <= second always
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    double os = __shfl_xor(s, o);
    s = os < s ? os : s;
  }
  if (lane == 0) {
    sh.red_best[w] = b;
    sh.red_idx[w] = bi;
    sh.red_second[w] = s;
  }
  __syncthreads();
  double gb = sh.red_best[0];
  int gi = sh.red_idx[0];
#pragma unroll
  for (int k = 1; k < SH::kNW; k++) {
    double ob = sh.red_best[k];
    int oi = sh.red_idx[k];
    bool take = (ob < gb) || (ob == gb && oi < gi);
    gb = take ? ob : gb;
    gi = take ? oi : gi;
  }
  double gs = rpp::dinf();
#pragma unroll
  for (int k = 0; k < SH::kNW; k++) {
    double c = (sh.red_idx[k] == gi) ? sh.red_second[k] : sh.red_best[k];
    gs = c < gs ? c : gs;
  }
  gbest = gb;
  gidx = gi;
  gsecond = gs;
  __syncthreads();
}

// ---------------------------------------------------------------------------
// Streaming argmin of dx*dx+dy*dy over x[0..n), y[0..n): each wave streams one
// contiguous quarter with 16-byte loads (lane -> 2 adjacent nodes), UNROLL
// independent load pairs in flight per lane.  Arrays are +inf padded.
template <class SH>
__device__ __forceinline__ void scan_nearest(const double* __restrict__ x, const double* __restrict__ y, int n,
                                             double qx, double qy, SH& sh, int& ni, double& gbest, double& gsecond) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = roundup_i((n + SH::kNW - 1) / SH::kNW, WAVE_STRIDE);
  const int ws = w * per;
  const int we = ws + per;
  double best = rpp::dinf(), second = rpp::dinf();
  int bidx = 0x7fffffff;
  for (int base = ws; base < we && base < n; base += WAVE_STRIDE) {
    v2d xv[UNROLL], yv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 128 + lane * 2;
      xv[u] = stream2(x + i0);
      yv[u] = stream2(y + i0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 128 + lane * 2;
      {
        double dx = xv[u].x - qx, dy = yv[u].x - qy;
        double d = dx * dx + dy * dy;
        bool lt = d < best;
        second = lt ? best : (d < second ? d : second);
        bidx = lt ? i0 : bidx;
        best = lt ? d : best;
      }
      {
        double dx = xv[u].y - qx, dy = yv[u].y - qy;
        double d = dx * dx + dy * dy;
        bool lt = d < best;
        second = lt ? best : (d < second ? d : second);
        bidx = lt ? i0 + 1 : bidx;
        best = lt ? d : best;
      }
    }
  }
  block_argmin(best, bidx, second, sh, gbest, ni, gsecond);
}

// ---------------------------------------------------------------------------
// Streaming threshold scan: indices with dx*dx+dy*dy <= thr are appended, in
// ascending order, to hits[] (wave w owns the slots starting at its range start;
// in-wave order by ballot prefix).  Returns the total; sh.wave_cnt/wave_start
// describe the four segments.
template <class SH>
__device__ __forceinline__ int scan_hits(const double* __restrict__ x, const double* __restrict__ y, int n, double qx,
                                         double qy, double thr, int32_t* __restrict__ hits, SH& sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = roundup_i((n + SH::kNW - 1) / SH::kNW, WAVE_STRIDE);
  const int ws = w * per;
  const int we = ws + per;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int cnt = 0;
  for (int base = ws; base < we && base < n; base += WAVE_STRIDE) {
    v2d xv[UNROLL], yv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 128 + lane * 2;
      xv[u] = stream2(x + i0);
      yv[u] = stream2(y + i0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 128 + lane * 2;
      double dx0 = xv[u].x - qx, dy0 = yv[u].x - qy;
      double dx1 = xv[u].y - qx, dy1 = yv[u].y - qy;
      bool h0 = (dx0 * dx0 + dy0 * dy0) <= thr;
      bool h1 = (dx1 * dx1 + dy1 * dy1) <= thr;
      uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
      if ((m0 | m1) != 0ull) {
        int pos = cnt + __popcll(m0 & lt_mask) + __popcll(m1 & lt_mask);
        if (h0) hits[ws + pos] = i0;
        if (h1) hits[ws + pos + (h0 ? 1 : 0)] = i0 + 1;
        cnt += __popcll(m0) + __popcll(m1);
      }
    }
  }
  if (lane == 0) {
    sh.wave_cnt[w] = cnt;
    sh.wave_start[w] = ws;
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int k = 0; k < SH::kNW; k++) total += sh.wave_cnt[k];
  return total;
}

// ---------------------------------------------------------------------------
// The same two passes over the f32 mirror xf[], yf[] (8 bytes per node; lane -> 4 adjacent nodes).  Distances from
// the mirror differ from the true ones by less than Ctx::f32_m, so the callers treat the results as candidates:
// the nearest index is final only if the runner-up is more than 2m further (distance metric), threshold hits are
// re-tested from the f64 coordinates.  See scan2f in rrt_star_v2_body.inc for the bound.
typedef float v4f_k __attribute__((ext_vector_type(4)));
constexpr int WAVE_STRIDE_F = 256 * UNROLL;
template <class SH>
__device__ __forceinline__ void scan_nearest_f32(const float* __restrict__ xf, const float* __restrict__ yf, int n,
                                                 float qx, float qy, SH& sh, int& ni, double& gbest, double& gsecond) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = roundup_i((n + SH::kNW - 1) / SH::kNW, WAVE_STRIDE_F);
  const int ws = w * per;
  const int we = ws + per;
  float best = __builtin_inff(), second = __builtin_inff();
  int bidx = 0x7fffffff;
  for (int base = ws; base < we && base < n; base += WAVE_STRIDE_F) {
    v4f_k xv[UNROLL], yv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 256 + lane * 4;
      xv[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f_k*>(xf + i0));
      yv[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f_k*>(yf + i0));
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 256 + lane * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float dx = xv[u][j] - qx, dy = yv[u][j] - qy;
        const float d = __builtin_fmaf(dx, dx, dy * dy);
        const bool lt = d < best;
        second = lt ? best : (d < second ? d : second);
        bidx = lt ? i0 + j : bidx;
        best = lt ? d : best;
      }
    }
  }
  block_argmin((double)best, bidx, (double)second, sh, gbest, ni, gsecond);
}
template <class SH>
__device__ __forceinline__ int scan_hits_f32(const float* __restrict__ xf, const float* __restrict__ yf, int n, float qx,
                                             float qy, float thr, int32_t* __restrict__ hits, SH& sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = roundup_i((n + SH::kNW - 1) / SH::kNW, WAVE_STRIDE_F);
  const int ws = w * per;
  const int we = ws + per;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int cnt = 0;
  for (int base = ws; base < we && base < n; base += WAVE_STRIDE_F) {
    v4f_k xv[UNROLL], yv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 256 + lane * 4;
      xv[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f_k*>(xf + i0));
      yv[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f_k*>(yf + i0));
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 256 + lane * 4;
      bool hh[4];
      uint64_t mm[4], any = 0ull;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float dx = xv[u][j] - qx, dy = yv[u][j] - qy;
        hh[j] = __builtin_fmaf(dx, dx, dy * dy) <= thr;
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
  if (lane == 0) {
    sh.wave_cnt[w] = cnt;
    sh.wave_start[w] = ws;
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int k = 0; k < SH::kNW; k++) total += sh.wave_cnt[k];
  return total;
}

// ---------------------------------------------------------------------------
// Fused pass: ONE stream over x[0..n), y[0..n) serves two queries -- the near-ball
// threshold scan about (qx,qy) of THIS iteration (rrt_04:1335-1337) and the
// nearest-node argmin about (sx,sy), the sample of the NEXT iteration
// (rrt_04:1198-1200).  The next sample does not depend on the tree (the RNG stream
// is consumed in the same order), and the node appended at the end of this
// iteration is folded into the argmin afterwards, so every node is read once per
// iteration instead of twice.
__device__ __forceinline__ int scan_fused(const double* __restrict__ x, const double* __restrict__ y, int n, double qx,
                                          double qy, double thr, double sx, double sy, int32_t* __restrict__ hits,
                                          Sh& sh, int& ni, double& gbest, double& gsecond) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = roundup_i((n + NW - 1) / NW, WAVE_STRIDE);
  const int ws = w * per;
  const int we = ws + per;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int cnt = 0;
  double best = rpp::dinf(), second = rpp::dinf();
  int bidx = 0x7fffffff;
  for (int base = ws; base < we && base < n; base += WAVE_STRIDE) {
    v2d xv[UNROLL], yv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 128 + lane * 2;
      xv[u] = stream2(x + i0);
      yv[u] = stream2(y + i0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int i0 = base + u * 128 + lane * 2;
      {
        double dx = xv[u].x - sx, dy = yv[u].x - sy;
        double d = dx * dx + dy * dy;
        bool lt = d < best;
        second = lt ? best : (d < second ? d : second);
        bidx = lt ? i0 : bidx;
        best = lt ? d : best;
      }
      {
        double dx = xv[u].y - sx, dy = yv[u].y - sy;
        double d = dx * dx + dy * dy;
        bool lt = d < best;
        second = lt ? best : (d < second ? d : second);
        bidx = lt ? i0 + 1 : bidx;
        best = lt ? d : best;
      }
      double dx0 = xv[u].x - qx, dy0 = yv[u].x - qy;
      double dx1 = xv[u].y - qx, dy1 = yv[u].y - qy;
      bool h0 = (dx0 * dx0 + dy0 * dy0) <= thr;
      bool h1 = (dx1 * dx1 + dy1 * dy1) <= thr;
      uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
      if ((m0 | m1) != 0ull) {
        int pos = cnt + __popcll(m0 & lt_mask) + __popcll(m1 & lt_mask);
        if (h0) hits[ws + pos] = i0;
        if (h1) hits[ws + pos + (h0 ? 1 : 0)] = i0 + 1;
        cnt += __popcll(m0) + __popcll(m1);
      }
    }
  }
  if (lane == 0) {
    sh.wave_cnt[w] = cnt;
    sh.wave_start[w] = ws;
  }
  block_argmin(best, bidx, second, sh, gbest, ni, gsecond);  // contains the barriers that publish wave_cnt
  int total = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) total += sh.wave_cnt[k];
  return total;
}

// h-th hit of the (virtual) concatenated, ascending list
template <class SH>
__device__ __forceinline__ int hit_at(const int32_t* hits, const SH& sh, int h) {
  int k = 0;
#pragma unroll
  for (int j = 0; j < SH::kNW - 1; j++) {
    if (k == j && h >= sh.wave_cnt[j]) {
      h -= sh.wave_cnt[j];
      k = j + 1;
    }
  }
  return hits[sh.wave_start[k] + h];
}
// position of that h-th hit inside hits[] (the raw-list walk of rewire stores a slot number there)
template <class SH>
__device__ __forceinline__ int hit_loc(const SH& sh, int h) {
  int k = 0;
#pragma unroll
  for (int j = 0; j < SH::kNW - 1; j++) {
    if (k == j && h >= sh.wave_cnt[j]) {
      h -= sh.wave_cnt[j];
      k = j + 1;
    }
  }
  return sh.wave_start[k] + h;
}

// ---------------------------------------------------------------------------
// Exact re-check + the `[lst.index(v) for v in lst if v <= thr]` idiom
// (rrt_04:1337, :1288-1291): every hit whose EXACT value passes reports the first
// index holding an equal value, so the distinct indices are the hits that are the
// first holder of their value, ascending.  mode 0: value = dx**2+dy**2 about
// (qx,qy); mode 1: value = math.hypot(x-qx, y-qy).
// Out: sh.uidx/uval[0..nu), sh.nu, sh.nvalid (= len of the reference's list).
template <class SH>
__device__ __forceinline__ void exact_dedup(const double* __restrict__ x, const double* __restrict__ y, double qx,
                                            double qy, double thr_exact, int mode, const int32_t* hits, int kraw,
                                            SH& sh) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) {
    sh.nu = 0;
    sh.nvalid = 0;
  }
  __syncthreads();
  for (int base = 0; base < kraw; base += SH::kTPB) {
    const int h = base + tid;
    int idx = -1;
    double v = 0.0;
    bool valid = false;
    if (h < kraw) {
      idx = hit_at(hits, sh, h);
      double dx = x[idx] - qx, dy = y[idx] - qy;
      v = (mode == 0) ? rpp::py_d2(dx, dy) : rpp::py_hypot(dx, dy);
      valid = v <= thr_exact;
    }
    const int nu = sh.nu;
    // "an equal value is held by an earlier entry": both searches run the same trip count on every lane with
    // broadcast LDS reads and no early exit -- throughput-bound instead of a dependent read-compare-branch per step
    // (near sets of hundreds of nodes made these loops the longest part of the candidate phase)
    bool dupe = false;
#pragma unroll 4
    for (int u = 0; u < nu; u++) dupe |= (sh.uval[u] == v);
    const bool cand = valid && !dupe;
    sh.cval[tid] = cand ? v : rpp::b2d(0x7ff8000000000000ULL);   // NaN: never equal
    sh.cflag[tid] = cand ? 1 : 0;
    __syncthreads();
    const int nchunk = (kraw - base) < SH::kTPB ? (kraw - base) : SH::kTPB;
    bool earlier = false;
#pragma unroll 4
    for (int t = 0; t < nchunk; t++) earlier |= (t < tid) & (sh.cval[t] == v);
    const bool first = cand && !earlier;
    uint64_t mf = __ballot(first), mv = __ballot(valid);
    // per-wave counts through red_idx (free between reductions)
    if (lane == 0) {
      sh.red_idx[w] = __popcll(mf);
      atomicAdd(&sh.nvalid, __popcll(mv));
    }
    __syncthreads();
    int off = nu;
#pragma unroll
    for (int k = 0; k < SH::kNW; k++)
      if (k < w) off += sh.red_idx[k];
    int tot = nu;
#pragma unroll
    for (int k = 0; k < SH::kNW; k++) tot += sh.red_idx[k];
    if (first) {
      const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      int p = off + __popcll(mf & lt_mask);
      if (p < SH::kNU) {
        sh.uval[p] = v;
        sh.uidx[p] = idx;
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (tot > SH::kNU) {
        sh.overflow = 1;
        tot = SH::kNU;
      }
      sh.nu = tot;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Steer + collision for `ne` edges.  kind 0: node uidx[e] -> (tx,ty) (choose_parent
// :1265, goal :1295); kind 1: (tx,ty) -> node uidx[e] (rewire :1359).
// Out per e: sh.uex/uey (edge end), sh.usafe (no collision AND end inside play area).
__device__ __forceinline__ void eval_edges(const Ctx& c, const double* __restrict__ x, const double* __restrict__ y,
                                           int ne, int kind, double tx, double ty, Sh& sh) {
  const int tid = threadIdx.x;
  for (int base = 0; base < ne; base += EB) {
    const int nb = (ne - base) < EB ? (ne - base) : EB;
    if (tid < nb) {
      const int u = sh.uidx[base + tid];
      const double ux = x[u], uy = y[u];
      if (kind == 0)
        rpp::steer(&sh.edge[tid], ux, uy, tx, ty, rpp::dinf(), c.res);
      else
        rpp::steer(&sh.edge[tid], tx, ty, ux, uy, rpp::dinf(), c.res);
      sh.ecoll[tid] = 0;
    }
    __syncthreads();
    for (int p = tid; p < nb * c.m; p += TPB) {
      const int e = p / c.m, k = p - e * c.m;
      if (rpp::edge_hits_obstacle(sh.edge[e], sh.ox[k], sh.oy[k], sh.othr[k])) sh.ecoll[e] = 1;
    }
    __syncthreads();
    if (tid < nb) {
      const rpp::Edge& e = sh.edge[tid];
      const int s = (!sh.ecoll[tid]) && rpp::in_play_area(c.has_play, c.play_area, e.ex, e.ey);
      if (kind == 0) {
        sh.uex[base + tid] = e.ex;
        sh.uey[base + tid] = e.ey;
        sh.usafe[base + tid] = s;
      } else {  // backward edges recomputed from the true new-node position: refresh bits 1,2 only
        const int s2 = (e.ex == e.tx) && (e.ey == e.ty);
        sh.usafe[base + tid] = (sh.usafe[base + tid] & 1) | (s << 1) | (s2 << 2);
      }
    }
    __syncthreads();
  }
}

// block-wide minimum of an int (block-uniform result)
__device__ __forceinline__ int block_min_int(int v, Sh& sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    int ov = __shfl_xor(v, o);
    v = ov < v ? ov : v;
  }
  if (lane == 0) sh.red_idx[w] = v;
  __syncthreads();
  int r = sh.red_idx[0];
#pragma unroll
  for (int k = 1; k < NW; k++) r = sh.red_idx[k] < r ? sh.red_idx[k] : r;
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------
// Both directions of every candidate edge in one go (choose_parent's steer(node -> new), rrt_04:1265, and
// rewire's steer(new -> node), :1359, speculating that the new node keeps the extension's coordinates, which
// it does whenever the winning edge snaps onto it).  The four waves split the serial libm chain: wave w handles
// direction w>>1 and evaluates cos (w even) or sin (w odd) of the edge angle; lane = candidate.
// Out per candidate e: uex/uey = end of the forward edge, uaux = hypot(new - node) (= hypot(node - new)),
// usafe bit0 = forward edge safe, bit1 = backward edge safe, bit2 = backward edge ends exactly on the node.
__device__ __forceinline__ void eval_edges_dual(const Ctx& c, const double* __restrict__ x, const double* __restrict__ y,
                                                int nu, double nx, double ny, Sh& sh) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kind = w >> 1, trig = w & 1;
  for (int base = 0; base < nu; base += EBD) {
    const int nb = (nu - base) < EBD ? (nu - base) : EBD;
    const bool act = lane < nb;
    rpp::Edge& E = sh.edge[kind * EBD + (lane & (EBD - 1))];
    if (act) {
      const int u = sh.uidx[base + lane];
      const double ux = x[u], uy = y[u];
      const double fx = kind ? nx : ux, fy = kind ? ny : uy, tx = kind ? ux : nx, ty = kind ? uy : ny;
      const double dx = tx - fx, dy = ty - fy;
      const double d = rpp::py_hypot(dx, dy);
      const double theta = rpp_glibc_atan2(dy, dx);
      if (trig == 0) {
        E.sx = c.res * rpp_glibc_cos(theta);
        E.fx = fx; E.fy = fy; E.tx = tx; E.ty = ty;
        E.n_expand = (int)__builtin_floor(d / c.res);   // extend_length = inf -> d (:1094-1097)
        if (kind == 0) sh.uaux[base + lane] = d;
      } else {
        E.sy = c.res * rpp_glibc_sin(theta);
      }
    }
    __syncthreads();
    if (act && trig == 0) {
      double px = E.fx, py = E.fy;
      const double sx = E.sx, sy = E.sy;
      for (int i = 0; i < E.n_expand; i++) {
        px += sx;
        py += sy;
      }
      const int snapped = rpp::py_hypot(E.tx - px, E.ty - py) <= c.res;
      E.ex = snapped ? E.tx : px;
      E.ey = snapped ? E.ty : py;
      E.snapped = snapped;
      sh.ecoll[kind * EBD + lane] = 0;
    }
    __syncthreads();
    for (int p = tid; p < 2 * nb * c.m; p += TPB) {
      const int q = p / c.m, k = p - q * c.m;
      const int kk = q / nb, e = q - kk * nb;
      const int slot = kk * EBD + e;
      if (rpp::edge_hits_obstacle(sh.edge[slot], sh.ox[k], sh.oy[k], sh.othr[k])) sh.ecoll[slot] = 1;
    }
    __syncthreads();
    if (tid < nb) {
      const rpp::Edge& f = sh.edge[tid];
      const rpp::Edge& b = sh.edge[EBD + tid];
      sh.uex[base + tid] = f.ex;
      sh.uey[base + tid] = f.ey;
      const int s0 = (!sh.ecoll[tid]) && rpp::in_play_area(c.has_play, c.play_area, f.ex, f.ey);
      const int s1 = (!sh.ecoll[EBD + tid]) && rpp::in_play_area(c.has_play, c.play_area, b.ex, b.ey);
      const int s2 = (b.ex == b.tx) && (b.ey == b.ty);
      sh.usafe[base + tid] = s0 | (s1 << 1) | (s2 << 2);
    }
    __syncthreads();
  }
}

// propagate_cost_to_leaves (rrt_04:1379-1384) as a level-synchronous sweep over the child lists by the whole
// workgroup (values are order independent).  Frontiers live in LDS (Sh::cval); returns the
// number of nodes rewritten, or -1 when a level outgrew the LDS frontier (caller falls back to the one-lane walk).
__device__ __forceinline__ int propagate_bfs(const double* __restrict__ x, const double* __restrict__ y,
                                             double* __restrict__ cost, const int32_t* __restrict__ first_child,
                                             const int32_t* __restrict__ next_sib, int root, Sh& sh) {
  int32_t* A = reinterpret_cast<int32_t*>(sh.cval);
  int32_t* Bf = A + FCAP;
  const int tid = threadIdx.x;
  if (tid == 0) {
    A[0] = root;
    sh.fa = 1;
    sh.fb = 0;
    sh.fover = 0;
    sh.fcount = 0;
  }
  __syncthreads();
  for (;;) {
    const int na = sh.fa;
    if (na == 0) break;
    int cnt = 0;
    for (int i = tid; i < na; i += TPB) {
      const int p = A[i];
      const double cp = cost[p], xp = x[p], yp = y[p];
      int ch = first_child[p];
      while (ch >= 0) {
        const double xc = x[ch], yc = y[ch];
        const int nxs = next_sib[ch];
        cost[ch] = cp + rpp::py_hypot(xc - xp, yc - yp);   // calc_new_cost :1375-1377
        const int slot = atomicAdd(&sh.fb, 1);
        if (slot < FCAP)
          Bf[slot] = ch;
        else
          sh.fover = 1;
        ch = nxs;
        cnt++;
      }
    }
    if (cnt) atomicAdd(&sh.fcount, cnt);
    __syncthreads();
    if (sh.fover) return -1;
    int32_t* t = A;
    A = Bf;
    Bf = t;
    __syncthreads();
    if (tid == 0) {
      sh.fa = sh.fb;
      sh.fb = 0;
    }
    __syncthreads();
  }
  return sh.fcount;
}

// first minimum of sh.uaux[0..ne) (list order), block-wide
__device__ __forceinline__ void first_min(int ne, Sh& sh, double& mn, int& sel) {
  double best = rpp::dinf(), second = rpp::dinf();
  int bidx = 0x7fffffff;
  for (int e = threadIdx.x; e < ne; e += TPB) {
    double d = sh.uaux[e];
    if (d < best) {
      best = d;
      bidx = e;
    }
  }
  double gs;
  block_argmin(best, bidx, second, sh, mn, sel, gs);
}

// propagate_cost_to_leaves (rrt_04:1379-1384) from `root`, one lane, explicit stack.
__device__ inline int propagate(double* __restrict__ x, double* __restrict__ y, double* __restrict__ cost,
                                const int32_t* __restrict__ first_child, const int32_t* __restrict__ next_sib,
                                int32_t* __restrict__ stack, int root) {
  int sp = 0, count = 0;
  stack[sp++] = root;
  while (sp) {
    const int p = stack[--sp];
    const double cp = cost[p], xp = x[p], yp = y[p];
    for (int ch = first_child[p]; ch >= 0; ch = next_sib[ch]) {
      cost[ch] = cp + rpp::py_hypot(x[ch] - xp, y[ch] - yp);  // calc_new_cost :1375-1377
      stack[sp++] = ch;
      count++;
    }
  }
  return count;
}

__device__ inline void link_child(int32_t* parent, int32_t* first_child, int32_t* next_sib, int32_t* prev_sib, int ch,
                                  int par) {
  parent[ch] = par;
  prev_sib[ch] = -1;
  if (par >= 0) {
    const int f = first_child[par];
    next_sib[ch] = f;
    if (f >= 0) prev_sib[f] = ch;
    first_child[par] = ch;
  } else {
    next_sib[ch] = -1;
  }
}
__device__ inline void unlink_child(int32_t* parent, int32_t* first_child, int32_t* next_sib, int32_t* prev_sib,
                                    int ch) {
  const int par = parent[ch];
  if (par < 0) return;
  const int pv = prev_sib[ch], nx = next_sib[ch];
  if (pv >= 0)
    next_sib[pv] = nx;
  else
    first_child[par] = nx;
  if (nx >= 0) prev_sib[nx] = pv;
}

// rewire (rrt_04:1357-1373) walked over the RAW near_inds list, one visit at a time, from the first visit of candidate
// slot `es0` on.  near_inds holds `dist_list.index(d)` (:1337): nodes at equal distance collapse onto the first of them
// and that index is listed again.  As long as no node has changed coordinates in this iteration a later visit of an index
// is void (costs only fall while the list is walked, the edge is the same), so the candidate loop of the caller visits
// every distinct index once.  When a rewire MOVES a node (an unsnapped steer, :1105-1110 with :1372) that no longer
// holds: the moved node is steered to where it lies now on its next visit, and costs of its descendants -- recomputed
// from the new position (:1375-1384) -- may rise, so that an index visited in vain before qualifies later.  From the first
// such rewire on, every remaining entry of the raw list is therefore visited as the reference visits it, against the
// current coordinates and costs.  dist_list itself predates the loop: the entry -> index map is taken before anything
// moves (stored over hits[], as candidate slot numbers).  Rare (needs a distance tie in the near set and an inexact
// path_resolution); lane 0 walks, the whole workgroup builds the map.
__device__ inline void rewire_raw_walk(const Ctx& c, double* __restrict__ x, double* __restrict__ y,
                                       double* __restrict__ cost, int32_t* parent, int32_t* first_child,
                                       int32_t* next_sib, int32_t* prev_sib, int32_t* hits, int32_t* stack, int kraw,
                                       int nu, double nx, double ny, double r2, double wx, double wy, double wcost,
                                       int newidx, int es0, Sh& sh) {
  const int tid = threadIdx.x;
  for (int base = 0; base < kraw; base += TPB) {
    const int h = base + tid;
    int loc = -1;
    double v = 0.0;
    bool valid = false;
    if (h < kraw) {
      loc = hit_loc(sh, h);
      const int idx = hits[loc];
      v = rpp::py_d2(x[idx] - nx, y[idx] - ny);
      valid = v <= r2;   // :1337
    }
    int slot = -1;
    for (int cb = 0; cb < nu; cb += TPB) {
      __syncthreads();
      if (cb + tid < nu) {
        const int u = sh.uidx[cb + tid];
        sh.cval[tid] = rpp::py_d2(x[u] - nx, y[u] - ny);
      }
      __syncthreads();
      const int nc = (nu - cb) < TPB ? (nu - cb) : TPB;
      if (valid && slot < 0) {
        for (int t = 0; t < nc; t++)
          if (sh.cval[t] == v) {   // dist_list.index(d): the first holder of the value
            slot = cb + t;
            break;
          }
      }
    }
    if (h < kraw) hits[loc] = slot;   // -1: outside the ball (the pass over-collects)
  }
  __syncthreads();
  if (tid == 0) {
    bool started = false;
    int moved = 0;
    for (int h = 0; h < kraw; h++) {
      const int e = hits[hit_loc(sh, h)];
      if (e < 0) continue;
      if (!started) {
        if (e != es0) continue;
        started = true;
      }
      const int u = sh.uidx[e];
      const double ux = x[u], uy = y[u];
      const double ec = wcost + rpp::py_hypot(ux - wx, uy - wy);   // calc_new_cost(new_node, near_node) :1362
      if (!(cost[u] > ec)) continue;                               // improved_cost :1366 (strict)
      rpp::steer(&sh.edge[0], wx, wy, ux, uy, rpp::dinf(), c.res);   // :1359
      bool ok = rpp::in_play_area(c.has_play, c.play_area, sh.edge[0].ex, sh.edge[0].ey);
      for (int k = 0; k < c.m && ok; k++)
        if (rpp::edge_hits_obstacle(sh.edge[0], sh.ox[k], sh.oy[k], sh.othr[k])) ok = false;
      if (!ok) continue;
      unlink_child(parent, first_child, next_sib, prev_sib, u);   // :1369-1371 (also when it already hangs under newidx)
      if (sh.edge[0].ex != ux || sh.edge[0].ey != uy) {
        x[u] = sh.edge[0].ex;   // node_list[i] = edge_node :1372
        y[u] = sh.edge[0].ey;
        moved = 1;
      }
      cost[u] = ec;
      link_child(parent, first_child, next_sib, prev_sib, u, newidx);
      sh.flag++;
      sh.sel += propagate(x, y, cost, first_child, next_sib, stack, u);   // :1373
    }
    if (moved) sh.nvalid = 1;
  }
  __syncthreads();
}

// generate_final_course (rrt_04:1117-1125) + get_path_length (:1391-1399), lane 0.
__device__ inline void write_path(const Ctx& c, Inst* I, const double* x, const double* y, const int32_t* parent,
                                  int inst, int gi) {
  double* out = c.path_xy + (int64_t)inst * c.path_cap * 2;
  int np = 0, trunc = 0;
  double px = I->goal[0], py = I->goal[1], len = 0.0;
  out[0] = px;
  out[1] = py;
  np = 1;
  int nd = gi;
  for (;;) {
    const double cx = x[nd], cy = y[nd];
    len += rpp::py_hypot(cx - px, cy - py);
    if (np < c.path_cap) {
      out[2 * np] = cx;
      out[2 * np + 1] = cy;
    } else {
      trunc = 1;
    }
    np++;
    px = cx;
    py = cy;
    if (parent[nd] < 0) break;
    nd = parent[nd];
  }
  I->path_n = np;
  I->goal_node = gi;
  I->status |= 2 | (trunc ? 8 : 0);
  c.results[inst].path_cost = len;
}

// search_best_goal_node (rrt_04:1284-1312): returns the goal node index or -1 (block-uniform).
__device__ __forceinline__ int best_goal_node(const Ctx& c, Inst* I, const double* x, const double* y,
                                              const double* cost, int32_t* hits, int n, Sh& sh, int64_t& eu,
                                              int64_t& er) {
  const double gx = I->goal[0], gy = I->goal[1];
  const double thr = c.expand_dis * c.expand_dis * (1.0 + FILTER_EPS);
  const int kraw = scan_hits(x, y, n, gx, gy, thr, hits, sh);
  exact_dedup(x, y, gx, gy, c.expand_dis, 1, hits, kraw, sh);
  const int nu = sh.nu;
  eu += nu;
  er += sh.nvalid;
  if (nu == 0) return -1;
  eval_edges(c, x, y, nu, 0, gx, gy, sh);
  for (int e = threadIdx.x; e < nu; e += TPB)
    sh.uaux[e] = sh.usafe[e] ? cost[sh.uidx[e]] + sh.uval[e] : rpp::dinf();  // cost + calc_dist_to_goal :1303-1305
  __syncthreads();
  double mn;
  int sel;
  first_min(nu, sh, mn, sel);
  if (!(mn < rpp::dinf())) return -1;
  return sh.uidx[sel];
}

// get_random_node / get_random_node_sobol (rrt_04:1132-1153), lane 0; result in sh.rx, sh.ry
template <class SH>
__device__ __forceinline__ void draw_sample(const Ctx& c, SH& sh, rpp::Sobol& sob, double gx, double gy) {
  double rx, ry;
  if (rpp::mt_randint_0_100(&sh.rng) > c.goal_sample_rate) {
    if (c.sampler == 1) {   // rrt_04:1142-1153, rrt_02:1077-1089
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

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(TPB, 4) void rrt_plan_kernel(Ctx c, int iters) {
  __shared__ Sh sh;
  const int inst = c.inst_map ? c.inst_map[blockIdx.x] : blockIdx.x;
  const int tid = threadIdx.x;
  Inst* I = c.inst + inst;
  if (I->status & 1) return;  // done
  const int64_t off = (int64_t)inst * c.stride;
  double* __restrict__ x = c.x + off;
  double* __restrict__ y = c.y + off;
  double* __restrict__ cost = c.cost + off;
  int32_t* parent = c.parent + off;
  int32_t* first_child = c.first_child + off;
  int32_t* next_sib = c.next_sib + off;
  int32_t* prev_sib = c.prev_sib + off;
  int32_t* hits = c.hits + off;
  int32_t* stack = c.stack + off;

  // stage per-instance state and the obstacle tile in LDS
  for (int i = tid; i < 624; i += TPB) sh.rng.mt[i] = I->rng.mt[i];
  for (int i = tid; i < c.m; i += TPB) {
    sh.ox[i] = c.ox[i];
    sh.oy[i] = c.oy[i];
    sh.othr[i] = c.othr[i];
  }
  if (tid == 0) {
    sh.rng.pos = I->rng.pos;
    sh.overflow = 0;
  }
  __syncthreads();
  int n = I->n, it = I->it;
  const double gx = I->goal[0], gy = I->goal[1];
  rpp::Sobol sob = I->sobol;
  int64_t s_iter = 0, s_eu = 0, s_er = 0, s_nh = 0, s_nu = 0, s_rw = 0, s_pr = 0, s_sn = 0, s_ab = 0, s_ex = 0;
  int done = 0;
  int have_sample = 0, have_nearest = 0, pf_ni = 0;   // block-uniform prefetch state (never carried across launches)
  double pf_best = 0.0, pf_second = 0.0;
  int64_t s_ab2 = 0;
  PH_DECL

  for (int step = 0; step < iters && it < c.max_iter && !done; step++, it++) {
    s_iter++;
    PH(15);
    // ---------------- sample (lane 0) rrt_04:1132-1153 (already drawn when the previous iteration prefetched it)
    if (!have_sample) {
      if (tid == 0) draw_sample(c, sh, sob, gx, gy);
      __syncthreads();
    }
    const double rx = sh.rx, ry = sh.ry;
    have_sample = 0;
    PH(0);

    // ---------------- nearest :1197-1202 (already known when the previous iteration's fused pass covered it)
    int ni;
    double gbest, gsecond;
    s_ab2 += 16 * (int64_t)n + 24 * (int64_t)c.m;
    s_ab += 24 * (int64_t)c.m;
    if (have_nearest) {
      ni = pf_ni;
      gbest = pf_best;
      gsecond = pf_second;
      have_nearest = 0;
    } else {
      scan_nearest(x, y, n, rx, ry, sh, ni, gbest, gsecond);
      s_sn += n;
      s_ab += 16 * (int64_t)n;
    }
    if (gbest != 0.0 && gsecond <= gbest * (1.0 + FILTER_EPS)) {
      // two candidates inside the filter margin: re-decide with the exact ** 2
      s_ex++;
      const int kraw = scan_hits(x, y, n, rx, ry, gbest * (1.0 + FILTER_EPS), hits, sh);
      double best = rpp::dinf(), second = rpp::dinf();
      int bidx = 0x7fffffff;
      for (int h = tid; h < kraw; h += TPB) {
        const int idx = hit_at(hits, sh, h);
        const double d = rpp::py_d2(x[idx] - rx, y[idx] - ry);
        if (d < best || (d == best && idx < bidx)) {
          best = d;
          bidx = idx;
        }
      }
      double gb2, gs2;
      block_argmin(best, bidx, second, sh, gb2, ni, gs2);
    }
    PH(1);

    // ---------------- steer + collision of the extension :1051-1059
    if (tid == 0) {
      rpp::steer(&sh.edge[0], x[ni], y[ni], rx, ry, c.expand_dis, c.res);
      sh.ecoll[0] = 0;
      sh.nx = sh.edge[0].ex;
      sh.ny = sh.edge[0].ey;
      sh.ncost = cost[ni] + rpp::py_hypot(sh.edge[0].ex - x[ni], sh.edge[0].ey - y[ni]);  // :1054-1056
      sh.flag = rpp::in_play_area(c.has_play, c.play_area, sh.edge[0].ex, sh.edge[0].ey) ? 1 : 0;
    }
    __syncthreads();
    PH(2);
    const double nx = sh.nx, ny = sh.ny;
    const int inplay = sh.flag;
    if (inplay) {
      s_eu++;
      s_er++;
      for (int k = tid; k < c.m; k += TPB)
        if (rpp::edge_hits_obstacle(sh.edge[0], sh.ox[k], sh.oy[k], sh.othr[k])) sh.ecoll[0] = 1;
    }
    __syncthreads();
    const int accepted = inplay && !sh.ecoll[0];
    int nnear = -1;
    int node_kind = 0;
    PH(3);

    if (accepted && c.algo == 0) {
      node_kind = 1;
      // ---- rrt_01:85-96
      if (tid == 0) {
        x[n] = nx;
        y[n] = ny;
        cost[n] = 0.0;
        first_child[n] = -1;
        link_child(parent, first_child, next_sib, prev_sib, n, ni);
      }
      n++;
      s_ab += 28;
      s_ab2 += 28;
      __syncthreads();
    }
    if (c.algo == 0) {
      // goal test on node_list[-1] (rrt_01:89-96)
      if (tid == 0) {
        const int last = n - 1;
        int ok = 0;
        if (rpp::py_hypot(x[last] - gx, y[last] - gy) <= c.expand_dis) {
          rpp::steer(&sh.edge[0], x[last], y[last], gx, gy, c.expand_dis, c.res);
          ok = 1;
        }
        sh.flag = ok;
        sh.ecoll[0] = 0;
      }
      __syncthreads();
      if (sh.flag) {
        s_eu++;
        s_er++;
        for (int k = tid; k < c.m; k += TPB)
          if (rpp::edge_hits_obstacle(sh.edge[0], sh.ox[k], sh.oy[k], sh.othr[k])) sh.ecoll[0] = 1;
        __syncthreads();
        if (!sh.ecoll[0]) {
          if (tid == 0) write_path(c, I, x, y, parent, inst, n - 1);
          done = 1;
        }
      }
      __syncthreads();
    }

    if (accepted && c.algo == 1) {
      // ---------------- find_near_nodes :1314-1338
      const double r2 = c.r2tab[n + 1];
      // Prefetch: draw the NEXT iteration's sample now (same RNG order: nothing below consumes the stream) and
      // let this pass also find its nearest node.  Only when the loop is known to run on (search_until_max_iter,
      // not the last iteration of this launch / of the plan), so the RNG state handed back is the reference's.
      const int do_pf = c.until_max && (step + 1 < iters) && (it + 1 < c.max_iter);
      const int n_scan = n;
      int kraw;
      if (do_pf) {
        if (tid == 0) draw_sample(c, sh, sob, gx, gy);
        __syncthreads();
        kraw = scan_fused(x, y, n, nx, ny, r2 * (1.0 + FILTER_EPS), sh.rx, sh.ry, hits, sh, pf_ni, pf_best, pf_second);
        have_sample = 1;
        have_nearest = 1;
      } else {
        kraw = scan_hits(x, y, n, nx, ny, r2 * (1.0 + FILTER_EPS), hits, sh);
      }
      s_sn += n;
      s_ab2 += 16 * (int64_t)n;
      s_ab += 16 * (int64_t)n;
      PH(4);
      exact_dedup(x, y, nx, ny, r2, 0, hits, kraw, sh);
      PH(5);
      const int nu = sh.nu;
      const int nvalid = sh.nvalid;
      nnear = nu;
      s_nh += nvalid;
      s_nu += nu;
      s_ab += 48 * (int64_t)nu + 28;
      s_ab2 += 48 * (int64_t)nu + 28;
      // ---------------- choose_parent :1242-1282 (+ speculative backward edges for rewire)
      int have = 0, sel = -1;
      double min_cost = rpp::dinf();
      if (nu > 0) {
        s_eu += nu;
        s_er += nvalid;
        eval_edges_dual(c, x, y, nu, nx, ny, sh);
        PH(6);
        for (int e = tid; e < nu; e += TPB) {
          const int u = sh.uidx[e];
          sh.uval[e] = (sh.usafe[e] & 1) ? cost[u] + sh.uaux[e] : rpp::dinf();  // near.cost + hypot(new - near) :1269
        }
        __syncthreads();
        {  // first minimum in list order (:1272-1278)
          double best = rpp::dinf(), second = rpp::dinf(), gs;
          int bidx = 0x7fffffff;
          for (int e = tid; e < nu; e += TPB) {
            const double d = sh.uval[e];
            if (d < best) {
              best = d;
              bidx = e;
            }
          }
          block_argmin(best, bidx, second, sh, min_cost, sel, gs);
        }
        have = min_cost < rpp::dinf();
        PH(7);
      }
      if (have) {
        // new_node = steer(node_list[min_ind], new_node); cost = min_cost  (:1279-1280)
        const double wx = sh.uex[sel], wy = sh.uey[sel], wcost = min_cost;
        const int min_ind = sh.uidx[sel];
        const int newidx = n;
        __syncthreads();
        // ---------------- rewire (before append) :1340-1373
        s_eu += nu;
        s_er += nvalid;
        if (wx != nx || wy != ny) {
          // the winning edge stopped short of the extension point: redo the backward edges from where it ended
          eval_edges(c, x, y, nu, 1, wx, wy, sh);
          for (int e = tid; e < nu; e += TPB) {
            const int u = sh.uidx[e];
            sh.uaux[e] = rpp::py_hypot(x[u] - wx, y[u] - wy);
          }
          __syncthreads();
        }
        for (int e = tid; e < nu; e += TPB) {
          sh.uval[e] = wcost + sh.uaux[e];   // edge_node.cost = new.cost + hypot(near - new) :1362
          sh.uex[e] = cost[sh.uidx[e]];       // current cost of the candidate (refreshed after every propagation)
        }
        if (tid == 0) {
          first_child[newidx] = -1;
          sh.flag = 0;    // rewires
          sh.sel = 0;     // propagated nodes
          sh.nvalid = 0;  // a node changed coordinates
        }
        __syncthreads();
        PH(8);
        // list order; later entries see costs updated by earlier successes (:1357-1373)
        int raw_from = -1;
        for (int e0 = 0; e0 < nu;) {
          int cand = 0x7fffffff;
          for (int e = e0 + tid; e < nu; e += TPB) {
            if ((sh.usafe[e] & 2) && sh.uex[e] > sh.uval[e]) {  // no_collision and improved_cost (strict) :1366-1368
              cand = e;
              break;
            }
          }
          const int es = block_min_int(cand, sh);
          if (es == 0x7fffffff) break;
          const int u = sh.uidx[es];
          if (!(sh.usafe[es] & 4) && nvalid > nu) {
            // this rewire moves its node and near_inds has repeated entries (distance ties, :1337): from here on the
            // raw list is walked visit by visit, as the reference does (rewire_raw_walk)
            raw_from = es;
            break;
          }
          if (tid == 0) {
            unlink_child(parent, first_child, next_sib, prev_sib, u);
            if (!(sh.usafe[es] & 4)) {
              // steer(new -> node) did not snap onto the node: node_list[i] = edge_node moves it (:1372)
              rpp::steer(&sh.edge[0], wx, wy, x[u], y[u], rpp::dinf(), c.res);
              x[u] = sh.edge[0].ex;
              y[u] = sh.edge[0].ey;
              sh.nvalid = 1;
            }
            cost[u] = sh.uval[es];
            link_child(parent, first_child, next_sib, prev_sib, u, newidx);
            sh.flag++;
          }
          __syncthreads();
          int np = propagate_bfs(x, y, cost, first_child, next_sib, u, sh);   // :1373
          if (np < 0) {
            if (tid == 0) sh.fcount = propagate(x, y, cost, first_child, next_sib, stack, u);
            __syncthreads();
            np = sh.fcount;
          }
          if (tid == 0) sh.sel += np;
          for (int e = es + 1 + tid; e < nu; e += TPB) sh.uex[e] = cost[sh.uidx[e]];
          __syncthreads();
          e0 = es + 1;
        }
        if (raw_from >= 0)
          rewire_raw_walk(c, x, y, cost, parent, first_child, next_sib, prev_sib, hits, stack, kraw, nu, nx, ny, r2, wx,
                          wy, wcost, newidx, raw_from, sh);
        if (tid == 0) {
          // append :1065
          x[newidx] = wx;
          y[newidx] = wy;
          cost[newidx] = wcost;
          link_child(parent, first_child, next_sib, prev_sib, newidx, min_ind);
        }
        __syncthreads();
        PH(9);
        s_rw += sh.flag;
        s_pr += sh.sel;
        if (sh.nvalid) have_nearest = 0;  // a rewired node changed coordinates: the prefetched argmin is stale
        n++;
        node_kind = 2;
      } else {
        node_kind = 1;
        // choose_parent returned None: append the extension as it is (:1066-1067)
        if (tid == 0) {
          x[n] = nx;
          y[n] = ny;
          cost[n] = sh.ncost;
          first_child[n] = -1;
          link_child(parent, first_child, next_sib, prev_sib, n, ni);
        }
        n++;
        __syncthreads();
      }
    }

    if (have_nearest) {
      // fold the node appended by this iteration (index n-1, not covered by the fused pass) into the argmin;
      // it has the highest index, so it wins strict improvements only (first-minimum rule, rrt_04:1200)
      const int last = n - 1;
      const double dxl = x[last] - sh.rx, dyl = y[last] - sh.ry;
      const double dl = dxl * dxl + dyl * dyl;
      if (dl < pf_best) {
        pf_second = pf_best;
        pf_best = dl;
        pf_ni = last;
      } else if (dl < pf_second) {
        pf_second = dl;
      }
    }
    if (inst == c.trace_inst && tid == 0) {
      c.tr_rx[it] = rx;
      c.tr_ry[it] = ry;
      c.tr_near[it] = ni;
      c.tr_nn[it] = nnear;
      c.tr_kind[it] = node_kind;
    }

    // ---------------- early exit :1072-1076
    PH(11);
    if (c.algo == 1 && !c.until_max) {
      s_sn += n;
      s_ab += 16 * (int64_t)n;
      s_ab2 += 16 * (int64_t)n;
      const int gi = best_goal_node(c, I, x, y, cost, hits, n, sh, s_eu, s_er);
      if (gi >= 0) {
        if (tid == 0) write_path(c, I, x, y, parent, inst, gi);
        done = 1;
      }
      __syncthreads();
    }
    if (sh.overflow) done = 1;
    PH(12);
  }

  // ---------------- after the loop :1078-1084
  if (!done && it >= c.max_iter) {
    if (c.algo == 1) {
      s_sn += n;
      s_ab += 16 * (int64_t)n;
      s_ab2 += 16 * (int64_t)n;
      const int gi = best_goal_node(c, I, x, y, cost, hits, n, sh, s_eu, s_er);
      if (gi >= 0 && tid == 0) write_path(c, I, x, y, parent, inst, gi);
      __syncthreads();
    }
    done = 1;
  }

  // write back state
  __syncthreads();
  for (int i = tid; i < 624; i += TPB) I->rng.mt[i] = sh.rng.mt[i];
  if (tid == 0) {
    I->rng.pos = sh.rng.pos;
    I->sobol = sob;
    I->n = n;
    I->it = it;
    if (done) I->status |= 1;
    if (sh.overflow == 1) I->status |= 4;
    if (sh.overflow == 2) I->status |= 16;
    I->iterations += s_iter;
    I->edges_unique += s_eu;
    I->edges_ref += s_er;
    I->near_hits += s_nh;
    I->near_unique += s_nu;
    I->rewires += s_rw;
    I->propagated += s_pr;
    I->scan_nodes += s_sn;
    I->alg_bytes += s_ab;
    I->exact_rescans += s_ex;
    I->alg_bytes2 += s_ab2;
    PH_STORE(I);
    c.results[inst].n_nodes = n;
    c.results[inst].status = I->status;
  }
}

// initialise one instance's arrays: +inf padding, root node (rrt_04:1043)
__global__ void rrt_init_kernel(Ctx c) {
  const int inst = c.inst_map ? c.inst_map[blockIdx.y] : blockIdx.y;
  const int64_t off = (int64_t)inst * c.stride;
  const double inf = rpp::dinf();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < c.stride; i += (int64_t)gridDim.x * blockDim.x) {
    c.x[off + i] = inf;
    c.y[off + i] = inf;
    if (c.xf) {
      c.xf[off + i] = __builtin_inff();
      c.yf[off + i] = __builtin_inff();
    }
    if (c.xq) c.xq[off + i] = 0xffffffffu;
  }
}
__global__ void rrt_root_kernel(Ctx c, int ninst) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= ninst) return;
  const int inst = c.inst_map ? c.inst_map[slot] : slot;
  const int64_t off = (int64_t)inst * c.stride;
  Inst* I = c.inst + inst;
  c.x[off] = I->start[0];
  c.y[off] = I->start[1];
  if (c.xf) {
    c.xf[off] = (float)I->start[0];
    c.yf[off] = (float)I->start[1];
  }
  if (c.xq) c.xq[off] = quant16(c, I->start[0], I->start[1]);
  // a start that lies exactly on the goal is the first such node (every goal sample then duplicates the ROOT)
  I->first_goal = (I->start[0] == I->goal[0] && I->start[1] == I->goal[1]) ? 0 : -1;
  I->goal_dups = 0;
  if (c.elen) c.elen[off] = 0.0;
  c.cost[off] = 0.0;
  c.parent[off] = -1;
  c.first_child[off] = -1;
  c.next_sib[off] = -1;
  c.prev_sib[off] = -1;
  I->n = 1;
  I->it = 0;
  I->status = 0;
  I->goal_node = -1;
  I->path_n = 0;
  I->sobol.index = 0;
  I->sobol.lastq[0] = I->sobol.lastq[1] = I->sobol.lastq[2] = 0;
  I->iterations = I->edges_unique = I->edges_ref = I->near_hits = I->near_unique = 0;
  I->rewires = I->propagated = I->scan_nodes = I->alg_bytes = I->exact_rescans = I->alg_bytes2 = 0;
  I->nu_max = I->f32_fallbacks = I->q16_fallbacks = I->rides = 0;
  for (int k = 0; k < 16; k++) I->phase[k] = 0;
  c.results[inst].path_cost = 0.0;
  c.results[inst].n_nodes = 1;
  c.results[inst].status = 0;
}

// Counters of a plan summed over the instances on the device (rrtx_stats): the host reads one 304-byte record instead of
// copying every Inst back (2.7 KB each: 44 MB for 16 384 instances, 10+ ms of pageable-memory copy per plan).
struct StatsAcc {
  long long sum[15];     // iterations, edges_unique, edges_ref, near_hits, near_unique, rewires, propagated, scan_nodes,
                         // alg_bytes, exact_rescans, alg_bytes2, n, f32_fallbacks, q16_fallbacks, rides
  long long nu_max;
  int32_t status_or, pad_;
  long long phase[16];
};
__global__ void stats_reduce_kernel(const Inst* inst, int ninst, StatsAcc* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  long long v[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, nm = 0, ph[16];
  int st = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) ph[k] = 0;
  if (i < ninst) {
    const Inst& I = inst[i];
    v[0] = I.iterations; v[1] = I.edges_unique; v[2] = I.edges_ref; v[3] = I.near_hits; v[4] = I.near_unique;
    v[5] = I.rewires; v[6] = I.propagated; v[7] = I.scan_nodes; v[8] = I.alg_bytes; v[9] = I.exact_rescans;
    v[10] = I.alg_bytes2; v[11] = I.n; v[12] = I.f32_fallbacks; v[13] = I.q16_fallbacks; v[14] = I.rides;
    nm = I.nu_max;
    st = I.status;
#pragma unroll
    for (int k = 0; k < 16; k++) ph[k] = I.phase[k];
  }
  // wave sums first (64 instances per atomic)
#pragma unroll
  for (int k = 0; k < 15; k++) {
    long long t = v[k];
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd((unsigned long long*)&out->sum[k], (unsigned long long)t);
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    long long t = ph[k];
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd((unsigned long long*)&out->phase[k], (unsigned long long)t);
  }
  for (int o = 32; o >= 1; o >>= 1) {
    const long long om = __shfl_xor(nm, o);
    nm = om > nm ? om : nm;
    st |= __shfl_xor(st, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax((long long*)&out->nu_max, nm);
    atomicOr(&out->status_or, st);
  }
}

// parity harness for the arithmetic replicas (rrtx_selftest_math)
__global__ void selftest_kernel(int op, const double* a, const double* b, double* o, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0.0;
  switch (op) {
    case 0: r = rpp::py_hypot(a[i], b[i]); break;
    case 1: r = rpp::py_sq(a[i]); break;
    case 2: r = rpp_glibc_sin(a[i]); break;
    case 3: r = rpp_glibc_cos(a[i]); break;
    case 4: r = rpp_glibc_atan2(a[i], b[i]); break;
    case 5: {
      rpp::Edge e;
      rpp::steer(&e, 0.0, 0.0, a[i], b[i], rpp::dinf(), 0.25);
      r = e.ex;
    } break;
    case 6: r = __builtin_sqrt(a[i]); break;
    case 7: r = a[i] / b[i]; break;
    case 8: r = rpp_glibc_acos(a[i]); break;
    case 9: r = rpp_glibc_asin(a[i]); break;
    case 10: {   // Reeds-Shepp steer (rpp_rs.h): (0, 0, 0) -> (a, b, a + b), curvature 1, step 0.2: checksum of the path
      double px[256], py[256], pyaw[256];
      rpp::RsResult R;
      rpp::rs_plan(0.0, 0.0, 0.0, a[i], b[i], a[i] + b[i], 1.0, 0.2, px, py, pyaw, 256, &R);
      r = (R.err || R.n == 0 || R.n > 256) ? (double)R.err - 1000.0 * R.n : px[R.n - 1] + py[R.n / 2] + pyaw[R.n - 1] + R.len[0];
    } break;
  }
  o[i] = r;
}

}  // namespace rppk
