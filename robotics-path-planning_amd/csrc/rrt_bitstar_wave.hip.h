// rrt_bitstar_wave.hip.h -- BIT* (rrt_08) on the GPU, one wave (64 lanes) per planning instance.
// Reference: 10_path_planning_01_rrt_08_batch_informed_rrt_star.py BITStar.plan :236-331 and everything it calls;
// rpp_bitstar.h holds the sequential restatement (host-tested against the reference's goldens) this kernel follows
// statement by statement.  What the reference does with Python containers per iteration is a handful of linear
// passes -- queue "best" values = full rebuild + sort (:439-474), expand_vertex over every sample (:476-501),
// list.remove, dict deletion, an A*-style sweep of the tree (:524-556) -- and those passes are what the lanes share:
//   * first-minimum / maximum reductions over the vertex and edge queues (cached g/h terms, see rpp_bitstar.h);
//   * expand_vertex as a ballot-ordered append (edges enter the queue in sample order, as the dict iteration does);
//   * order-preserving deletion (list.remove, del dict[k]) as a chunked shift;
//   * connect()'s point-sampled collision test (:359-383) with the first colliding point found by ballot;
//   * update_graph with O(1) open / closed membership flags, successors enumerated in adjacency (append) order;
//   * informed_sample: the MT19937 draws stay sequential (lane 0, stream order), coordinates / grid ids in parallel,
//     dictionary insertion with the reference's key semantics (existing keys keep their slot, last value wins).
// Control flow is wave-uniform: every branch depends on reduced / broadcast values only.
// Small per-vertex state lives in LDS (max_iter + 1 <= VL vertices); samples and the edge queue in the instance's
// global slab (lane-strided, coalesced).  Larger problems use the one-lane-per-instance kernel of rrt_bitstar.hip.h.
#pragma once
#include "rrt_bitstar.hip.h"

namespace rppb {

constexpr int VL = 128;    // vertices / vertex queue / tree edges held in LDS
constexpr int RB = 404;    // random numbers of one informed_sample batch (2 per sample, <= 201 samples)
constexpr int OB = 64;     // obstacles held in LDS

struct ShB {
  rpp::MT rng;
  double vid[VL], vg[VL], vf[VL], vpar[VL], vh[VL], vq[VL];
  int32_t vhasp[VL], vq_i[VL], te_a[VL], te_b[VL], open[VL], inopen[VL], inclosed[VL];
  double rnd[RB];
  double ox[OB], oy[OB], othr[OB];
};

__device__ __forceinline__ void wsync() { __syncthreads(); }   // one wave per workgroup: orders LDS / global traffic

// (value, index) -> smallest value, lowest index among equals; index 0x7fffffff when no lane had a candidate
__device__ __forceinline__ void w_argmin(double& v, int& i) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(i, o);
    const bool take = (ov < v) || (ov == v && oi < i);
    v = take ? ov : v;
    i = take ? oi : i;
  }
}
__device__ __forceinline__ double w_max(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double ov = __shfl_xor(v, o);
    v = ov > v ? ov : v;
  }
  return v;
}
// first j in [from, n) with a[j] == key, else -1 (wave-uniform result)
// (four chunks of 64 entries are requested together and examined in order: one round trip per 256 entries)
template <class T>
__device__ __forceinline__ int w_find(const T* a, int from, int n, T key) {
  const int lane = threadIdx.x;
  for (int base = from; base < n; base += 256) {
    T v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = base + 64 * u + lane;
      v[u] = a[j < n ? j : from];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = base + 64 * u + lane;
      const uint64_t m = __ballot(j < n && v[u] == key);
      if (m) return base + 64 * u + __ffsll((long long)m) - 1;
    }
  }
  return -1;
}
__device__ __forceinline__ int w_find_pair(const double* a, const double* b, int n, double ka, double kb) {
  const int lane = threadIdx.x;
  for (int base = 0; base < n; base += 256) {
    double va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = base + 64 * u + lane;
      va[u] = a[j < n ? j : 0];
      vb[u] = b[j < n ? j : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = base + 64 * u + lane;
      const uint64_t m = __ballot(j < n && va[u] == ka && vb[u] == kb);
      if (m) return base + 64 * u + __ffsll((long long)m) - 1;
    }
  }
  return -1;
}
// del self.samples[id]: the three sample columns shifted together
__device__ __forceinline__ void w_erase3(double* a, double* b, double* c3, int ri, int n) {
  const int lane = threadIdx.x;
  for (int base = ri; base + 1 < n; base += 256) {
    double ta[4], tb[4], tc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = base + u * 64 + lane;
      if (j + 1 < n) {
        ta[u] = a[j + 1];
        tb[u] = b[j + 1];
        tc[u] = c3[j + 1];
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = base + u * 64 + lane;
      if (j + 1 < n) {
        a[j] = ta[u];
        b[j] = tb[u];
        c3[j] = tc[u];
      }
    }
    __syncthreads();
  }
}
// list.remove at position ri of an array of length n (order preserving)
template <class T>
__device__ __forceinline__ void w_erase(T* a, int ri, int n) {
  const int lane = threadIdx.x;
  for (int base = ri; base + 1 < n; base += 256) {
    T t[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = base + u * 64 + lane;
      if (j + 1 < n) t[u] = a[j + 1];
    }
    wsync();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = base + u * 64 + lane;
      if (j + 1 < n) a[j] = t[u];
    }
    wsync();
  }
}

// Persistent waves and a device-side work queue ("more instances than workgroup slots and a device-side work
// queue"): the grid holds as many waves as the chip keeps resident, each pulls the next pending instance from an atomic
// counter when its own is finished, so a batch larger than the resident set has no per-set tail and run lengths that differ
// from instance to instance (BIT* depends on the start / goal pair) pack.  a.queue lists the pending instances (nullptr =
// 0 .. n_pending-1).  A launch is BOUNDED: an instance that has made a.trip_bound trips of plan()'s loop (:243) stores its
// whole state (LDS columns -> its slab, the scalars -> out_i / save_i) and is carried into the next launch (ST_CARRY in its
// Inst status), where a wave picks it up again; the host re-queues what is not done (rrtx_api.hip).
constexpr int ST_CARRY = 0x100;   // Inst::status only (never in a result record): state stored, to be resumed

#ifndef RRTX_BIT_EW
#define RRTX_BIT_EW 8
#endif
constexpr int EW = RRTX_BIT_EW;   // edge-queue entries per lane and round trip (scan, shift)
// edge_queue.remove: the five columns of the edge queue shifted together (one load / store round per 64 EW entries instead
// of five per 256)
__device__ __forceinline__ void w_erase_edge(double* a, double* b, int32_t* ai, double* dab, double* hb, int ri, int n) {
  const int lane = threadIdx.x;
  for (int base = ri; base + 1 < n; base += 64 * EW) {
    double ta[EW], tb[EW], td[EW], th[EW];
    int32_t ti[EW];
#pragma unroll
    for (int u = 0; u < EW; u++) {
      const int j = base + u * 64 + lane;
      if (j + 1 < n) {
        ta[u] = a[j + 1];
        tb[u] = b[j + 1];
        ti[u] = ai[j + 1];
        td[u] = dab[j + 1];
        th[u] = hb[j + 1];
      }
    }
    wsync();
#pragma unroll
    for (int u = 0; u < EW; u++) {
      const int j = base + u * 64 + lane;
      if (j + 1 < n) {
        a[j] = ta[u];
        b[j] = tb[u];
        ai[j] = ti[u];
        dab[j] = td[u];
        hb[j] = th[u];
      }
    }
    wsync();
  }
}

__global__ __launch_bounds__(64) void bitstar_wave_kernel(BitArgs a, rppk::Inst* inst, rppk::Result* results,
                                                          int n_inst) {
  __shared__ ShB sh;
  const int lane = threadIdx.x;
 for (;;) {
  int slot = 0;
  if (lane == 0) slot = atomicAdd(a.qhead, 1);
  slot = __builtin_amdgcn_readfirstlane(slot);
  if (slot >= a.n_pending) break;
  const int I = a.queue ? a.queue[slot] : slot;
  if (I < 0 || I >= n_inst) continue;
  const bool resume = (inst[I].status & ST_CARRY) != 0;
  const rpp::BitCfg c = a.cfg[I];
  double* d = a.dslab + (int64_t)I * DSLAB;
  int32_t* q = a.islab + (int64_t)I * ISLAB;
  double* sid = d; d += SC;
  double* sx = d; d += SC;
  double* sy = d; d += SC;
  double* lid = d; d += LC;   // raw batch of informed_sample: id, x, y per drawn sample
  double* lx = d; d += LC;
  double* ly = d; d += LC;
  double* g_vid = d; d += VC;
  double* g_vg = d; d += VC;
  double* g_vf = d; d += VC;
  double* g_vpar = d; d += VC;
  d += VC;   // vq (LDS here)
  double* eq_a = d; d += EC;
  double* eq_b = d; d += EC;
  double* path = d; d += 2LL * PC;
  double* eq_dab = d; d += EC;
  double* eq_hb = d; d += EC;
  int32_t* g_vhasp = q; q += 5LL * VC;
  int32_t* eq_ai = q;
  double* tr_a = (I == a.trace_inst) ? a.tr_a : nullptr;
  double* tr_b = (I == a.trace_inst) ? a.tr_b : nullptr;

  for (int i = lane; i < 624; i += 64) sh.rng.mt[i] = inst[I].rng.mt[i];
  for (int i = lane; i < c.m; i += 64) {
    sh.ox[i] = c.ox[i];
    sh.oy[i] = c.oy[i];
    sh.othr[i] = c.othr[i];
  }
  if (lane == 0) sh.rng.pos = inst[I].rng.pos;
  wsync();

  // wave-uniform scalars
  int ns = 0, nv = 0, nte = 0, nvq = 0, neq = 0, path_n = 0, tr_n = 0, error = 0, iterations = 0, found_goal = 0;
  int seen0 = 0;     // the queues have run dry once with iterations == 0 (see the hang test in the loop)
  int trips = 0, carry = 0;
  long guard = 0;
  double g_goal = rpp::dinf();
  const double inf = rpp::dinf();
  const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  double* g_vq = g_vpar + VC;                       // slab homes of the LDS columns (carry-over only)
  double* g_vh = eq_hb + EC;
  int32_t* g_te_a = g_vhasp + VC;
  int32_t* g_te_b = g_vhasp + 2LL * VC;
  int32_t* g_vq_i = eq_ai + EC;
  int32_t* sv = a.save_i + 8 * I;

  const double start_id = rpp::bit_id(c, c.start[0], c.start[1]), goal_id = rpp::bit_id(c, c.goal[0], c.goal[1]);
  if (!resume) {
    if (lane == 0) {
      sid[0] = goal_id;
      sx[0] = c.goal[0];
      sy[0] = c.goal[1];
      sh.vid[0] = start_id;
      sh.vg[0] = 0.0;
      sh.vf[0] = rpp::bit_dist(c, start_id, goal_id);
      sh.vh[0] = rpp::bit_dist(c, start_id, goal_id);
      sh.vhasp[0] = 0;
      sh.vpar[0] = -1.0;
    }
    ns = 1;
    nv = 1;
  } else {
    const int32_t* o = a.out_i + 8 * I;
    nv = o[0]; nte = o[1]; ns = o[2]; iterations = o[5]; tr_n = o[6]; found_goal = o[7];
    nvq = sv[0]; neq = sv[1]; seen0 = sv[4];
    guard = (long)(((uint64_t)(uint32_t)sv[3] << 32) | (uint64_t)(uint32_t)sv[2]);
    g_goal = a.out_g[I];
    for (int v = lane; v < VL; v += 64) {
      if (v < nv) {
        sh.vid[v] = g_vid[v]; sh.vg[v] = g_vg[v]; sh.vf[v] = g_vf[v]; sh.vpar[v] = g_vpar[v]; sh.vh[v] = g_vh[v];
        sh.vhasp[v] = g_vhasp[v];
      }
      if (v < nvq) {
        sh.vq[v] = g_vq[v];
        sh.vq_i[v] = g_vq_i[v];
      }
      if (v < nte) {
        sh.te_a[v] = g_te_a[v];
        sh.te_b[v] = g_te_b[v];
      }
    }
  }
  wsync();

  // informed_sample(m, cMax, ...) :397-420 followed by self.samples.update(...)
  auto informed_sample = [&](int mm, double c_max) {
    const int cnt = mm + 1;
    if (lane == 0)
      for (int i = 0; i < 2 * cnt; i++) sh.rnd[i] = rpp::mt_random(&sh.rng);   // stream order: a, b (or x, y) per sample
    wsync();
    for (int i = lane; i < cnt; i += 64) {
      double rx, ry;
      if (c_max < inf) {
        const double r0 = c_max / 2.0;
        const double r1 = __builtin_sqrt(rpp::py_sq(c_max) - c.c_min2) / 2.0;
        double aa = sh.rnd[2 * i], bb = sh.rnd[2 * i + 1];
        if (bb < aa) {
          const double t = aa;
          aa = bb;
          bb = t;
        }
        const double ang = 2 * 3.141592653589793 * aa / bb;
        const double s0 = bb * rpp_glibc_cos(ang), s1 = bb * rpp_glibc_sin(ang);
        const double t00 = c.rot[0] * r0, t01 = c.rot[1] * r1, t10 = c.rot[2] * r0, t11 = c.rot[3] * r1;
        rx = __builtin_fma(t00, s0, t01 * s1) + (c.start[0] + c.goal[0]) / 2.0;
        ry = __builtin_fma(t10, s0, t11 * s1) + (c.start[1] + c.goal[1]) / 2.0;
      } else {
        rx = c.rand_min + (c.rand_max - c.rand_min) * sh.rnd[2 * i];       // sample_free_space :433-437
        ry = c.rand_min + (c.rand_max - c.rand_min) * sh.rnd[2 * i + 1];
      }
      lid[i] = rpp::bit_id(c, rx, ry);
      lx[i] = rx;
      ly[i] = ry;
    }
    wsync();
    for (int i = lane; i < cnt; i += 64) sh.rnd[i] = lid[i];   // ids of the batch in LDS
    wsync();
    // does any drawn id repeat (inside the batch or against the samples already held)?
    int dup = 0;
    for (int i = lane; i < cnt; i += 64) {
      const double id = sh.rnd[i];
      for (int j = 0; j < i; j++)
        if (sh.rnd[j] == id) dup = 1;
    }
    for (int base = 0; base < ns; base += 64) {
      const int k = base + lane;
      if (k < ns) {
        const double id = sid[k];
        for (int j = 0; j < cnt; j++)
          if (sh.rnd[j] == id) dup = 1;
      }
    }
    const bool anydup = __ballot(dup) != 0ull;
    if (!anydup) {
      // all keys new: dict order = draw order
      if (ns + cnt > SC) {
        error = 2;
        return;
      }
      for (int i = lane; i < cnt; i += 64) {
        sid[ns + i] = sh.rnd[i];
        sx[ns + i] = lx[i];
        sy[ns + i] = ly[i];
      }
      ns += cnt;
    } else {
      // exact dict semantics, one key at a time: an existing key keeps its slot and takes the later value
      for (int i = 0; i < cnt; i++) {
        const double id = sh.rnd[i];
        int pos = w_find(sid, 0, ns, id);
        if (pos < 0) {
          if (ns >= SC) {
            error = 2;
            return;
          }
          pos = ns++;
          if (lane == 0) sid[pos] = id;
        }
        if (lane == 0) {
          sx[pos] = lx[i];
          sy[pos] = ly[i];
        }
        wsync();
      }
    }
    wsync();
  };

  if (!resume) informed_sample(200, g_goal);

  while (iterations < c.max_iter && error == 0) {
    if (trips++ >= a.trip_bound) {   // this launch's share is used up: store the state, another launch resumes here
      carry = 1;
      break;
    }
    if (++guard > 4000000) {
      error = 2;
      break;
    }
    // ---- setup_sample :209-234
    if (nvq == 0 && neq == 0) {
      if (iterations == 0) {
        // The reference adds samples only `if iterations != 0` (:215): while no edge has ever connected (a start walled in by
        // obstacles: every connect() fails and `continue`s past the iteration counter, :283) it comes back here with the
        // tree, the samples and the RNG exactly as they were the first time -- plan() then repeats the same round for ever.
        // The second arrival proves it; the instance ends at once instead of spinning up to the trip guard.
        if (seen0) {
          error = 3;
          break;
        }
        seen0 = 1;
      }
      if (iterations != 0) {
        int mm = 100;
        if (found_goal) {
          mm = 200;
          if (lane == 0) {
            sid[0] = goal_id;
            sx[0] = c.goal[0];
            sy[0] = c.goal[1];
          }
          ns = 1;
          wsync();
        }
        informed_sample(mm, g_goal);
        if (error) break;
      }
      // the vertex queue is empty here: it becomes the list of all tree vertices
      for (int v = lane; v < nv; v += 64) {
        sh.vq[v] = sh.vid[v];
        sh.vq_i[v] = v;
      }
      nvq = nv;
      wsync();
    }
    // ---- while best_vertex_queue_value() <= best_edge_queue_value(): expand_vertex(best_in_vertex_queue()) :249-251
    int emin_i = 0x7fffffff;
    bool fail = false;
    for (;;) {
      double bv = inf;
      int bvi = 0x7fffffff;
      for (int j = lane; j < nvq; j += 64) {
        const int vi = sh.vq_i[j];
        const double val = sh.vg[vi] + sh.vh[vi];
        if (val < bv) {
          bv = val;
          bvi = j;
        }
      }
      w_argmin(bv, bvi);
      double be = inf;
      emin_i = 0x7fffffff;
      if (neq) {
        double mx = -inf, mn = inf;
        int mi = 0x7fffffff;
        // EW queue entries per lane and round trip (3 EW independent loads in flight): with thousands of queued edges this
        // scan, once per popped edge and per expanded vertex, is what the slowest instances of a batch spend their time in
        for (int j0 = lane; j0 < neq; j0 += 64 * EW) {
          int ai[EW];
          double da[EW], hb[EW];
#pragma unroll
          for (int u = 0; u < EW; u++) {
            const int j = j0 + 64 * u;
            const int jj = j < neq ? j : j0;   // in range: the value is not used
            ai[u] = eq_ai[jj];
            da[u] = eq_dab[jj];
            hb[u] = eq_hb[jj];
          }
#pragma unroll
          for (int u = 0; u < EW; u++) {
            const int j = j0 + 64 * u;
            if (j < neq) {
              const double val = sh.vg[ai[u]] + da[u] + hb[u];
              mx = val > mx ? val : mx;   // values.sort(reverse=True)[0]: the MAXIMUM (:452-453)
              if (val < mn) {
                mn = val;
                mi = j;
              }
            }
          }
        }
        be = w_max(mx);
        w_argmin(mn, mi);
        emin_i = mi;
      }
      if (!(bv <= be)) break;
      if (nvq == 0) {
        error = 1;   // IndexError in best_in_vertex_queue
        fail = true;
        break;
      }
      if (bvi == 0x7fffffff) bvi = 0;
      const double vid = sh.vq[bvi];
      const int vidx = sh.vq_i[bvi];
      wsync();
      w_erase(sh.vq, bvi, nvq);     // vertex_queue.remove(vid)
      w_erase(sh.vq_i, bvi, nvq);
      nvq--;
      const double d_sv = rpp::bit_dist(c, start_id, vid);
      double cx, cy;
      rpp::bit_coord(c, vid, &cx, &cy);
      for (int base0 = 0; base0 < ns && !fail; base0 += 256) {   // samples.items() in dict order; RAW sample coordinates (:485-488)
       // four chunks of 64 samples requested together, appended chunk by chunk in order
       double p_id[4], p_x[4], p_y[4];
#pragma unroll
       for (int u = 0; u < 4; u++) {
         const int k = base0 + 64 * u + lane;
         const int kk = k < ns ? k : 0;
         p_id[u] = sid[kk];
         p_x[u] = sx[kk];
         p_y[u] = sy[kk];
       }
#pragma unroll
       for (int u = 0; u < 4; u++) {
        const int base = base0 + 64 * u;
        if (base >= ns || fail) continue;
        const int k = base + lane;
        bool pred = false;
        double sidk = 0.0, d_sg = 0.0, d_vs = 0.0;
        if (k < ns) {
          sidk = p_id[u];
          if (rpp::bit_norm(p_x[u] - cx, p_y[u] - cy) <= 2.0 && sidk != vid) {
            d_sg = rpp::bit_dist(c, sidk, goal_id);
            d_vs = rpp::bit_dist(c, vid, sidk);
            const double est = d_sv + d_sg + d_vs;
            pred = est < g_goal;
          }
        }
        const uint64_t m = __ballot(pred);
        if (m) {
          const int add = __popcll(m);
          if (neq + add > EC) {
            error = 2;
            fail = true;
            continue;
          }
          if (pred) {
            const int pos = neq + __popcll(m & below);
            eq_a[pos] = vid;
            eq_b[pos] = sidk;
            eq_ai[pos] = vidx;
            eq_dab[pos] = d_vs;
            eq_hb[pos] = d_sg;
          }
          neq += add;
        }
       }
      }
      wsync();
      if (fail) break;
    }
    if (fail) break;
    // ---- bestEdge = best_in_edge_queue(); edge_queue.remove(bestEdge) :253-255
    if (neq == 0) {
      error = 1;
      break;
    }
    const int bi = (emin_i == 0x7fffffff) ? 0 : emin_i;
    const double ea = eq_a[bi], eb = eq_b[bi];
    const int ea_i = eq_ai[bi];
    const double dab = eq_dab[bi], hb = eq_hb[bi];
    if (tr_a && tr_n < a.tr_cap && lane == 0) {
      tr_a[tr_n] = ea;
      tr_b[tr_n] = eb;
    }
    tr_n++;
    {
      int ri = w_find_pair(eq_a, eq_b, neq, ea, eb);
      if (ri < 0) ri = 0;
      wsync();
      w_erase_edge(eq_a, eq_b, eq_ai, eq_dab, eq_hb, ri, neq);
      neq--;
    }
    const int v0 = ea_i;
    const double est_v = sh.vg[v0] + dab + hb;
    const double est_e = rpp::bit_dist(c, start_id, ea) + dab + hb;
    const double act_e = sh.vg[v0] + dab;
    if (est_v < g_goal && est_e < g_goal && act_e < g_goal) {   // f1 and f2 and f3 :270-273
      double fx, fy, tx, ty;
      rpp::bit_coord(c, ea, &fx, &fy);
      rpp::bit_coord(c, eb, &tx, &ty);
      // connect :359-374
      const int steps = (int)(rpp::bit_dist(c, rpp::bit_id(c, fx, fy), rpp::bit_id(c, tx, ty)) * 10);
      const double last_edge = rpp::bit_id(c, tx, ty);
      const double stepx = steps > 1 ? (tx - fx) / (steps - 1) : 0.0, stepy = steps > 1 ? (ty - fy) / (steps - 1) : 0.0;
      auto point = [&](int i, double* px, double* py) {
        if (steps > 1 && i == steps - 1) {
          *px = tx;
          *py = ty;
        } else {
          *px = (stepx == 0.0 && steps > 1) ? ((double)i / (steps - 1)) * (tx - fx) + fx : i * stepx + fx;
          *py = (stepy == 0.0 && steps > 1) ? ((double)i / (steps - 1)) * (ty - fy) + fy : i * stepy + fy;
        }
      };
      int npth = steps > 0 ? steps : 0;
      for (int base = 0; base < steps; base += 64) {
        const int i = base + lane;
        bool col = false;
        if (i < steps) {
          double px, py;
          point(i, &px, &py);
          for (int k = 0; k < c.m; k++) {
            const double dx = sh.ox[k] - px, dy = sh.oy[k] - py;
            if (dx * dx + dy * dy <= sh.othr[k]) col = true;
          }
        }
        const uint64_t m = __ballot(col);
        if (m) {
          npth = base + __ffsll((long long)m) - 1;
          break;
        }
      }
      if (npth == 0) continue;   // path None or empty: no iteration count (:283-284)
      double lx2, ly2;
      point(npth - 1, &lx2, &ly2);
      const double next_id = rpp::bit_id(c, lx2, ly2);
      if (w_find(sh.vid, 0, nv, next_id) >= 0) continue;   // :291-292
      {   // del self.samples[next_id]
        const int di = w_find(sid, 0, ns, next_id);
        if (di >= 0) {
          wsync();
          w_erase3(sid, sx, sy, di, ns);
          ns--;
        }
      }
      if (nv >= VL || nvq >= VL || nte >= VL) {
        error = 2;
        break;
      }
      const int vn = nv++;
      const double gs = rpp::bit_dist(c, ea, next_id);
      if (lane == 0) {
        sh.vid[vn] = next_id;
        sh.vhasp[vn] = 0;
        sh.vpar[vn] = -1.0;
        sh.vh[vn] = rpp::bit_dist(c, next_id, goal_id);
        sh.vq_i[nvq] = vn;
        sh.vq[nvq] = next_id;
        sh.te_a[nte] = v0;   // tree.add_edge :62-66
        sh.te_b[nte] = vn;
        sh.vg[vn] = gs + sh.vg[v0];
        sh.vf[vn] = gs + rpp::bit_dist(c, next_id, goal_id);
      }
      nvq++;
      nte++;
      if (next_id == goal_id || ea == goal_id) found_goal = 1;   // :300-303 (bestEdge rebound to (e0, next) :289)
      wsync();
      if (next_id == goal_id) g_goal = sh.vg[vn];
      // ---- update_graph :524-556
      {
        for (int v = lane; v < nv; v += 64) {
          sh.inopen[v] = 0;
          sh.inclosed[v] = 0;
        }
        if (lane == 0) {
          sh.open[0] = 0;
        }
        wsync();
        if (lane == 0) sh.inopen[0] = 1;
        int no = 1;
        wsync();
        while (no) {
          double mv = inf;
          int mj = 0x7fffffff;
          for (int j = lane; j < no; j += 64) {
            const double val = sh.vf[sh.open[j]];
            if (val < mv) {
              mv = val;
              mj = j;
            }
          }
          w_argmin(mv, mj);   // min(openSet, key=f): first minimum
          const int bj = (mj == 0x7fffffff) ? 0 : mj;
          const int cur = sh.open[bj];
          wsync();
          w_erase(sh.open, bj, no);
          no--;
          if (lane == 0) sh.inopen[cur] = 0;
          wsync();
          const double cur_id = sh.vid[cur];
          if (cur_id == goal_id) break;
          if (lane == 0) sh.inclosed[cur] = 1;
          wsync();
          for (int base = 0; base < nte; base += 64) {   // tree.vertices[cur] in append order
            const int e = base + lane;
            uint64_t m = __ballot(e < nte && (sh.te_a[e] == cur || sh.te_b[e] == cur));
            while (m) {
              const int e2 = base + __ffsll((long long)m) - 1;
              m &= m - 1;
              const int su = (sh.te_a[e2] == cur) ? sh.te_b[e2] : sh.te_a[e2];
              if (sh.inclosed[su]) continue;
              const double su_id = sh.vid[su];
              const double gsc = sh.vg[cur] + rpp::bit_dist(c, cur_id, su_id);
              if (!sh.inopen[su]) {
                if (lane == 0) {
                  sh.open[no] = su;
                  sh.inopen[su] = 1;
                }
                no++;
              } else if (gsc >= sh.vg[su]) {
                continue;
              }
              wsync();
              if (lane == 0) {
                sh.vg[su] = gsc;
                sh.vf[su] = gsc + rpp::bit_dist(c, su_id, goal_id);
                sh.vpar[su] = cur_id;
                sh.vhasp[su] = 1;
              }
              if (su_id == goal_id) g_goal = gsc;
              wsync();
            }
          }
        }
      }
      // ---- remove_queue(lastEdge, bestEdge) :349-357 (iterates the list it mutates)
      if (sh.vg[vn] + 0.0 >= g_goal) {
        int i = 0;
        for (;;) {
          const int p = w_find(eq_b, i, neq, next_id);
          if (p < 0) break;
          i = p + 1;
          const int ri = w_find_pair(eq_a, eq_b, neq, last_edge, next_id);
          if (ri >= 0) {
            wsync();
            w_erase_edge(eq_a, eq_b, eq_ai, eq_dab, eq_hb, ri, neq);
            neq--;
          }
        }
      }
    } else {   // "Nothing good" :322-325
      neq = 0;
      nvq = 0;
    }
    iterations++;
  }

  if (carry) {
    // ---- carried into the next launch: LDS columns -> slab, scalars -> out_i / save_i, RNG -> Inst
    wsync();
    for (int v = lane; v < VL; v += 64) {
      if (v < nv) {
        g_vid[v] = sh.vid[v]; g_vg[v] = sh.vg[v]; g_vf[v] = sh.vf[v]; g_vpar[v] = sh.vpar[v]; g_vh[v] = sh.vh[v];
        g_vhasp[v] = sh.vhasp[v];
      }
      if (v < nvq) {
        g_vq[v] = sh.vq[v];
        g_vq_i[v] = sh.vq_i[v];
      }
      if (v < nte) {
        g_te_a[v] = sh.te_a[v];
        g_te_b[v] = sh.te_b[v];
      }
    }
    for (int i = lane; i < 624; i += 64) inst[I].rng.mt[i] = sh.rng.mt[i];
    if (lane == 0) {
      inst[I].rng.pos = sh.rng.pos;
      int32_t* o = a.out_i + 8 * I;
      o[0] = nv; o[1] = nte; o[2] = ns; o[3] = 0; o[4] = 0; o[5] = iterations; o[6] = tr_n; o[7] = found_goal;
      sv[0] = nvq; sv[1] = neq; sv[2] = (int32_t)(uint32_t)((uint64_t)guard & 0xffffffffu);
      sv[3] = (int32_t)(uint32_t)((uint64_t)guard >> 32); sv[4] = seen0;
      a.out_g[I] = g_goal;
      inst[I].n = nv;
      inst[I].it = iterations;
      inst[I].status = ST_CARRY;
      results[I].path_cost = g_goal;
      results[I].n_nodes = nv;
      results[I].status = 0;
    }
    wsync();
    continue;   // next pending instance
  }
  // ---- find_final_path :333-347
  if (!error) {
    int np = 0;
    bool ok = true;
    auto push = [&](double x0, double x1) {
      if (np < PC && lane == 0) {
        path[2 * np] = x0;
        path[2 * np + 1] = x1;
      }
      np++;
    };
    push(c.goal[0], c.goal[1]);
    double cur = goal_id;
    int hops = 0;
    while (cur != start_id) {
      double cx, cy;
      rpp::bit_coord(c, cur, &cx, &cy);
      push(cx, cy);
      const int vi = w_find(sh.vid, 0, nv, cur);
      if (vi < 0 || !sh.vhasp[vi] || ++hops > VC + 2) {
        ok = false;   // KeyError: "cannot find Path" -> []
        break;
      }
      cur = sh.vpar[vi];
    }
    if (ok) {
      push(c.start[0], c.start[1]);
      if (np > PC) {
        error = 2;
      } else {
        wsync();
        if (lane == 0) {
          for (int i = 0; i < np / 2; i++) {   // plan[::-1]
            const int j = np - 1 - i;
            const double t0 = path[2 * i], t1 = path[2 * i + 1];
            path[2 * i] = path[2 * j];
            path[2 * i + 1] = path[2 * j + 1];
            path[2 * j] = t0;
            path[2 * j + 1] = t1;
          }
        }
        path_n = np;
      }
    }
  }
  wsync();
  // ---- results: tree columns back to the slab (rrtx_get_tree reads them), counters, RNG state
  for (int v = lane; v < nv; v += 64) {
    g_vid[v] = sh.vid[v];
    g_vg[v] = sh.vg[v];
    g_vf[v] = sh.vf[v];
    g_vpar[v] = sh.vpar[v];
    g_vhasp[v] = sh.vhasp[v];
  }
  for (int i = lane; i < 624; i += 64) inst[I].rng.mt[i] = sh.rng.mt[i];
  if (lane == 0) {
    inst[I].rng.pos = sh.rng.pos;
    int32_t* o = a.out_i + 8 * I;
    o[0] = nv; o[1] = nte; o[2] = ns; o[3] = path_n; o[4] = error; o[5] = iterations; o[6] = tr_n;
    o[7] = found_goal;
    a.out_g[I] = g_goal;
    inst[I].n = nv;
    inst[I].it = iterations;
    inst[I].iterations = iterations;
    inst[I].edges_unique = tr_n;
    inst[I].edges_ref = tr_n;
    inst[I].status = 1 | (path_n > 0 ? 2 : 0) | (error >= 2 ? 4 : 0) | (error == 3 ? 64 : 0);   // DONE, PATH, OVERFLOW, REF_HANGS
    results[I].path_cost = g_goal;
    results[I].n_nodes = nv;
    results[I].status = inst[I].status;
  }
  wsync();   // the LDS columns are reused by the next instance of the queue
 }
}

}  // namespace rppb
