// rrt_bitstar.hip.h -- BIT* (rrt_08) on the GPU: instance-parallel, one lane per planning instance.
// Reference: 10_path_planning_01_rrt_08_batch_informed_rrt_star.py BITStar.plan :236-331 (see rpp_bitstar.h).
// BIT* is a small sequential queue-driven search (<= a few hundred samples, <= maxIter vertices per instance) whose
// results depend on container order; SURVEY.md 8(a) B2 / 8(e): it parallelises across instances only.  This first
// version runs the sequential core (rpp::bitstar_plan, the same source the host unit test pins to the reference's
// goldens) on one lane per instance; per-instance state lives in global memory slabs.
#pragma once
#include "rpp_bitstar.h"
#include "rrt_kernels.hip.h"

namespace rppb {

constexpr int SC = 4096;     // samples (an instance that needs more ends with RRTX_ST_OVERFLOW: a walled-in start resamples without bound)
constexpr int LC = 512;      // samples of one batch
constexpr int VC = 1024;     // tree vertices / vertex queue / tree edges
constexpr int EC = 32768;    // edge queue
constexpr int PC = 2048;     // path points

struct BitArgs {
  rpp::BitCfg* cfg;          // per instance (start / goal / rotation may differ)
  double* dslab;             // per instance: doubles
  int32_t* islab;            // per instance: ints
  int32_t* out_i;            // per instance: nv, nte, ns, path_n, error, iterations, tr_n, found_goal
  double* out_g;             // per instance: g_goal
  double *tr_a, *tr_b;       // optional trace of instance trace_inst
  int32_t tr_cap, trace_inst;
  // wave kernel: device-side work queue and launch bound (rrt_bitstar_wave.hip.h)
  const int32_t* queue;      // pending instance ids of this launch (nullptr = 0 .. n_pending-1)
  int32_t* qhead;            // atomic counter: next queue slot
  int32_t n_pending, trip_bound;
  int32_t* save_i;           // per instance: nvq, neq, guard lo / hi, seen0 of a carried instance
};
// (the arrays rrtx_get_tree / rrtx_get_path read back keep their offsets; the cached columns follow them)
constexpr int64_t DSLAB = 3LL * SC + 3LL * LC + 5LL * VC + 2LL * EC + 2LL * PC + 2LL * EC + VC;
constexpr int64_t ISLAB = 5LL * VC + EC + VC;

__global__ void bitstar_kernel(BitArgs a, rppk::Inst* inst, rppk::Result* results, int n_inst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inst) return;
  double* d = a.dslab + (int64_t)i * DSLAB;
  int32_t* q = a.islab + (int64_t)i * ISLAB;
  rpp::BitState s;
  s.sid = d; d += SC; s.sx = d; d += SC; s.sy = d; d += SC;
  s.lid = d; d += LC; s.lx = d; d += LC; s.ly = d; d += LC;
  s.vid = d; d += VC; s.vg = d; d += VC; s.vf = d; d += VC; s.vpar = d; d += VC; s.vq = d; d += VC;
  s.eq_a = d; d += EC; s.eq_b = d; d += EC;
  s.path = d; d += 2LL * PC;
  s.eq_dab = d; d += EC; s.eq_hb = d; d += EC; s.vh = d;
  s.vhasp = q; q += VC; s.te_a = q; q += VC; s.te_b = q; q += VC; s.open = q; q += VC; s.closed = q; q += VC;
  s.eq_ai = q; q += EC; s.vq_i = q;
  s.scap = SC; s.lcap = LC; s.vcap = VC; s.tecap = VC; s.vqcap = VC; s.eqcap = EC; s.path_cap = PC;
  s.tr_a = (i == a.trace_inst) ? a.tr_a : nullptr;
  s.tr_b = (i == a.trace_inst) ? a.tr_b : nullptr;
  s.tr_cap = a.tr_cap;
  rpp::bitstar_plan(a.cfg[i], s, &inst[i].rng);
  int32_t* o = a.out_i + 8 * i;
  o[0] = s.nv; o[1] = s.nte; o[2] = s.ns; o[3] = s.path_n; o[4] = s.error; o[5] = s.iterations; o[6] = s.tr_n;
  o[7] = s.found_goal;
  a.out_g[i] = s.g_goal;
  inst[i].n = s.nv;
  inst[i].it = s.iterations;
  inst[i].iterations = s.iterations;
  inst[i].edges_unique = s.tr_n;
  inst[i].edges_ref = s.tr_n;
  inst[i].status = 1 | (s.path_n > 0 ? 2 : 0) | (s.error >= 2 ? 4 : 0) | (s.error == 3 ? 64 : 0);   // DONE, PATH, OVERFLOW, REF_HANGS
  results[i].path_cost = s.g_goal;
  results[i].n_nodes = s.nv;
  results[i].status = inst[i].status;
}

}  // namespace rppb
