// path_smooth.hip.h -- batched path_smoothing (rrt_04:1447-1479), the step every reference driver runs right after
// planning() (rrt_04:1558-1559): random shortcutting of the returned polyline, drawing from the same MT19937 stream.
// One 64-lane wave per polyline.  The polyline, its segment lengths d[i] (math.hypot, :1396 / :1408) and their running
// sums pre[i] (the `le += d` of get_path_length / get_target_point, accumulated in index order so every partial sum
// is the one the reference holds) live in LDS and are rebuilt only when a shortcut is accepted; get_target_point's
// scan (:1401-1421) is then "first i with pre[i] >= target" found by ballot, line_collision_check (:1423-1444, the
// INFINITE-line distance |a ox + b oy + c| / hypot(a, b) <= size, kept as is) runs one obstacle per lane.
// Python semantics kept: pickPoints.sort(), path[-1] when ti == -1 (:1418-1419, value unused because ti <= 0 is
// rejected), float division by zero -> the reference raises ZeroDivisionError -> status SM_ZERODIV here.
#pragma once
#include "rpp_core.h"

namespace rpps {

constexpr int PC = 512;       // polyline points held in LDS
constexpr int MOB = 256;      // obstacles held in LDS
enum { SM_OK = 0, SM_CAPACITY = 1, SM_ZERODIV = 2 };

struct SmoothArgs {
  const double* in_xy;        // job j: in_xy + j * in_stride * 2, in_n[j * in_n_stride] points (goal -> start order)
  int64_t in_stride;
  const int32_t* in_n;
  int64_t in_n_stride;        // in ints (lets in_n point into an array of structs)
  rpp::MT* rng;               // job j: (rpp::MT*)((char*)rng + j * rng_stride_bytes), advanced in place
  int64_t rng_stride_bytes;
  const double *ox, *oy, *osz;   // obstacle centres and SIZES (no robot radius here, :1441)
  int32_t m, max_iter;
  double* out_xy;             // job j: out_xy + j * out_stride * 2
  int64_t out_stride;
  int32_t *out_n, *status;
};

struct ShS {
  rpp::MT rng;
  double x[2][PC], y[2][PC], d[PC], pre[PC];
  double ox[MOB], oy[MOB], osz[MOB];
  double pick[2];
};

__global__ __launch_bounds__(64) void smooth_kernel(SmoothArgs a, int n_jobs) {
  __shared__ ShS sh;
  const int job = blockIdx.x, lane = threadIdx.x;
  if (job >= n_jobs) return;
  rpp::MT* grng = reinterpret_cast<rpp::MT*>(reinterpret_cast<char*>(a.rng) + job * a.rng_stride_bytes);
  int n = a.in_n[job * a.in_n_stride];
  int status = SM_OK;
  if (n > PC || a.m > MOB) status = SM_CAPACITY;
  if (n < 2 || status) {   // nothing to smooth (or planning() returned None): the input is the output
    if (lane == 0) {
      a.out_n[job] = (status || n < 0) ? 0 : n;
      a.status[job] = status;
    }
    if (!status)
      for (int i = lane; i < n; i += 64) {
        a.out_xy[(job * a.out_stride + i) * 2] = a.in_xy[(job * a.in_stride + i) * 2];
        a.out_xy[(job * a.out_stride + i) * 2 + 1] = a.in_xy[(job * a.in_stride + i) * 2 + 1];
      }
    return;
  }
  for (int i = lane; i < 624; i += 64) sh.rng.mt[i] = grng->mt[i];
  if (lane == 0) sh.rng.pos = grng->pos;
  for (int k = lane; k < a.m; k += 64) {
    sh.ox[k] = a.ox[k];
    sh.oy[k] = a.oy[k];
    sh.osz[k] = a.osz[k];
  }
  for (int i = lane; i < n; i += 64) {
    sh.x[0][i] = a.in_xy[(job * a.in_stride + i) * 2];
    sh.y[0][i] = a.in_xy[(job * a.in_stride + i) * 2 + 1];
  }
  __syncthreads();
  int cur = 0;
  // segment lengths and their running sums of buffer `cur` (get_path_length :1391-1399)
  auto measure = [&]() {
    for (int i = lane; i < n - 1; i += 64)
      sh.d[i] = rpp::py_hypot(sh.x[cur][i + 1] - sh.x[cur][i], sh.y[cur][i + 1] - sh.y[cur][i]);
    __syncthreads();
    if (lane == 0) {
      double le = 0;
      for (int i = 0; i < n - 1; i++) {
        le += sh.d[i];
        sh.pre[i] = le;
      }
    }
    __syncthreads();
  };
  // get_target_point :1401-1421 -> (x, y, ti); false when the reference would divide by zero
  auto target = [&](double t, double* px, double* py, int* ti_out) -> bool {
    int hit = -1;
    for (int base = 0; base < n - 1; base += 64) {
      const int i = base + lane;
      const uint64_t mk = __ballot(i < n - 1 && sh.pre[i] >= t);
      if (mk) {
        hit = base + __ffsll((long long)mk) - 1;
        break;
      }
    }
    if (hit < 0) return false;              // loop fell through: lastPairLen == 0
    const double last = sh.d[hit];
    if (last == 0.0) return false;
    const double ratio = (sh.pre[hit] - t) / last;
    const int ti = hit - 1;
    const int ia = ti < 0 ? n - 1 : ti, ib = hit;
    *px = sh.x[cur][ia] + (sh.x[cur][ib] - sh.x[cur][ia]) * ratio;
    *py = sh.y[cur][ia] + (sh.y[cur][ib] - sh.y[cur][ia]) * ratio;
    *ti_out = ti;
    return true;
  };
  measure();
  double le = sh.pre[n - 2];
  for (int it = 0; it < a.max_iter; it++) {
    if (lane == 0) {
      sh.pick[0] = rpp::mt_uniform(&sh.rng, 0.0, le);   // :1453
      sh.pick[1] = rpp::mt_uniform(&sh.rng, 0.0, le);
    }
    __syncthreads();
    double t0 = sh.pick[0], t1 = sh.pick[1];
    __syncthreads();
    if (t1 < t0) {   // pickPoints.sort()
      const double t = t0;
      t0 = t1;
      t1 = t;
    }
    double fx, fy, sx, sy;
    int fi, si;
    if (!target(t0, &fx, &fy, &fi) || !target(t1, &sx, &sy, &si)) {
      status = SM_ZERODIV;
      break;
    }
    if (fi <= 0 || si <= 0) continue;
    if (si + 1 > n) continue;
    if (si == fi) continue;
    // line_collision_check :1423-1444
    const double la = sy - fy, lb = -(sx - fx), lc = sy * (sx - fx) - sx * (sy - fy);
    const double h = rpp::py_hypot(la, lb);
    if (a.m > 0 && h == 0.0) {
      status = SM_ZERODIV;
      break;
    }
    bool blocked = false;
    for (int base = 0; base < a.m; base += 64) {
      const int k = base + lane;
      bool hitk = false;
      if (k < a.m) {
        const double dd = rpp::dabs(la * sh.ox[k] + lb * sh.oy[k] + lc) / h;
        hitk = dd <= sh.osz[k];
      }
      if (__ballot(hitk)) {
        blocked = true;
        break;
      }
    }
    if (blocked) continue;
    // newPath = path[:fi+1] + [first] + [second] + path[si+1:]  :1471-1476
    const int tail = n - si - 1;
    const int nn = fi + 1 + 2 + tail;
    if (nn > PC) {
      status = SM_CAPACITY;
      break;
    }
    const int nxt = cur ^ 1;
    for (int i = lane; i <= fi; i += 64) {
      sh.x[nxt][i] = sh.x[cur][i];
      sh.y[nxt][i] = sh.y[cur][i];
    }
    if (lane == 0) {
      sh.x[nxt][fi + 1] = fx;
      sh.y[nxt][fi + 1] = fy;
      sh.x[nxt][fi + 2] = sx;
      sh.y[nxt][fi + 2] = sy;
    }
    for (int i = lane; i < tail; i += 64) {
      sh.x[nxt][fi + 3 + i] = sh.x[cur][si + 1 + i];
      sh.y[nxt][fi + 3 + i] = sh.y[cur][si + 1 + i];
    }
    __syncthreads();
    cur = nxt;
    n = nn;
    measure();
    le = sh.pre[n - 2];
  }
  __syncthreads();
  for (int i = lane; i < n && i < a.out_stride; i += 64) {
    a.out_xy[(job * a.out_stride + i) * 2] = sh.x[cur][i];
    a.out_xy[(job * a.out_stride + i) * 2 + 1] = sh.y[cur][i];
  }
  if (n > a.out_stride && !status) status = SM_CAPACITY;
  for (int i = lane; i < 624; i += 64) grng->mt[i] = sh.rng.mt[i];
  if (lane == 0) {
    grng->pos = sh.rng.pos;
    a.out_n[job] = n;
    a.status[job] = status;
  }
}

}  // namespace rpps
