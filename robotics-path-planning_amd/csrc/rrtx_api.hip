// rrtx_api.hip -- host side of the C ABI declared in include/rrtx.h.
// Owns device memory, uploads parameters / RNG state, launches the planner
// kernel in bounded chunks of iterations on the handle's HIP stream, and copies
// results back.  There is no CPU planning path in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the functions are resolved with dlsym at first use (no link-time dependency)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rrtx.h"
#include "rrt_kernels.hip.h"
#include "rrt_star_v2.hip.h"
#include "rrt_informed.hip.h"
#include "rrt_dubins.hip.h"
#include "rrt_rs.hip.h"
#include "rrt_bitstar.hip.h"
#include "rrt_bitstar_wave.hip.h"
#include "path_smooth.hip.h"

using rppk::Ctx;
using rppk::Inst;
using rppk::Result;

namespace {

// libm through volatile pointers: the compiler must not fold pow(x, 2.0) into x*x
// (glibc's pow is not correctly rounded and the reference's `**2` goes through it).
double (*volatile libm_pow)(double, double) = pow;
double (*volatile libm_log)(double) = log;
double (*volatile libm_sqrt)(double) = sqrt;
double (*volatile libm_sin)(double) = sin;
double (*volatile libm_cos)(double) = cos;
double (*volatile libm_atan2)(double, double) = atan2;
double (*volatile libm_acos)(double) = acos;
double (*volatile libm_asin)(double) = asin;

inline double py_sq_host(double x) {
  if (x == 0.0) return 0.0;
  return libm_pow(std::fabs(x), 2.0);
}

}  // namespace

// RCCL, opened with dlopen at first use (rrtx_rccl_*): no link-time dependency, nothing loaded by single-GPU users
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
RcclApi* rccl_api() {
  static RcclApi api;
  if (api.lib || !api.err.empty()) return &api;
  for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (api.lib) break;
  }
  if (!api.lib) {
    api.err = std::string("librccl.so not loadable: ") + (dlerror() ? dlerror() : "?");
    return &api;
  }
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
  api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) {
    api.err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
    api.lib = nullptr;
  }
  return &api;
}
}  // namespace

struct rrtx_handle {
  rrtx_params p;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  Ctx c;
  int64_t stride = 0;
  int n_inst = 0;
  int m = 0;
  bool planned = false;
  std::vector<Inst> host_inst;  // staging for seeds / starts before the first plan
  std::vector<double> obst;
  int trace_inst = -1;
  rrtx_stats stats;
  int64_t phase[16] = {0};
  std::string err;
  std::vector<void*> allocs;
  // one rrtx_plan in progress (rrtx_plan_begin / rrtx_plan_step; rrtx_plan = begin + steps until nothing is pending)
  struct Run {
    int stage = 0;   // 0 none, 1 RRT* iteration-kernel launches, 2 launches of the planner's main kernel
    std::chrono::steady_clock::time_point t0;
    double kms = 0.0, kms_main = -1.0;
    int64_t launches = 0, launches_main = 0, steps = 0, v2_done_it = 0;
    bool use_v2 = false, v2_f32 = false, bit_wave = false;
    int v2_tpb = 0;
    std::vector<Result> res;
    std::vector<int32_t> pending;   // BIT*: instances not finished yet (the device-side work queue of the next launch)
  } run;
  int32_t *bit_queue = nullptr, *bit_qhead = nullptr;   // BIT*: device copy of `pending`, queue head counter
  int bit_trip_bound = 20000;    // BIT*: trips of plan()'s loop per instance and launch (rrt_bitstar_wave.hip.h)
  Inst* d_inst0 = nullptr;       // device copy of host_inst (the staged start state of every instance)
  bool inst_dirty = true;        // host_inst changed since d_inst0 was written
  rppk::StatsAcc* d_acc = nullptr;
  // native RCCL gather of the result table (rrtx_rccl_*): communicator of this rank, world size, receive buffer
  void* rccl_comm = nullptr;
  int rccl_world = 0, rccl_rank = 0;
  Result* rccl_recv = nullptr;
  int chunk_iters = 32768;       // iterations per launch of the other planner kernels
  int32_t* inst_map = nullptr;   // device: instance ids of a partial re-plan (overflow retry)
  // pose planners: where an instance's edge polylines live -- the handle's pool (slab = instance), or a larger pool
  // allocated for instances that outgrew it (slab = position in that re-plan)
  struct PoolLoc {
    double *px = nullptr, *py = nullptr, *pyaw = nullptr;
    int64_t cap = 0, slab = 0;
  };
  std::vector<PoolLoc> pool_loc;
  // the enlarged pools of the overflow re-plan (x4, x16): kept on the handle and used again by later rrtx_plan calls
  // when they are large enough (slabs >= instances to re-plan), so repeated plans do not grow device memory
  struct BigPool {
    double *px = nullptr, *py = nullptr, *pyaw = nullptr;
    int64_t cap = 0;
    int slabs = 0;
  } big[2];
  int32_t* pool_slot = nullptr;  // device copy of the slab numbers of a re-plan
  int64_t stats_retried = 0;
  int informed_eager = 0;        // rrt_07 kernel: 1 = collision-test every near candidate (reference order), 0 = cheapest first
  int v2_chunk_iters = 131072;  // iterations per launch of the RRT* iteration kernel (rrt_star_v2_body.inc)
  double* cbest = nullptr;  // informed RRT*: best path length so far per instance (device)
  std::vector<rppi::InformedArgs> iargs;   // informed RRT*: per-instance rotation C / c_min**2 (centre filled at plan time)
  std::vector<double> icmin;               // informed RRT*: per-instance c_min as handed in
  rppi::InformedArgs* d_iargs = nullptr;
  rppd::DubArgs da;         // RRT*-Dubins device arrays
  rppb::BitArgs ba;         // BIT* device arrays
  std::vector<rpp::BitCfg> bcfg;
  // path smoothing on the planned paths (rrtx_smooth_planned)
  double* sm_osz = nullptr;   // obstacle sizes as given (no robot radius), device
  double* sm_xy = nullptr;    // [inst][sm_stride][2]
  int32_t *sm_n = nullptr, *sm_status = nullptr;
  int64_t sm_stride = 0;
  bool smoothed = false;
};

#define HIPCHK(h, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
      return RRTX_E_HIP;                                                                     \
    }                                                                                        \
  } while (0)

template <class T>
static int dalloc(rrtx_handle* h, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) {
    h->err = std::string("hipMalloc: ") + hipGetErrorString(e);
    return RRTX_E_HIP;
  }
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}


// ---- RRT* (rrt_04, search_until_max_iter): iteration-kernel launches ------------------------------------------------
// One pass of the latency-lean iteration kernel over `nblk` instances (c.inst_map selects them; nullptr = 0..nblk-1),
// in chunks of h->v2_chunk_iters iterations, workgroup shape tpb in {64, 128, 256}.
static int launch_rrt_star_v2_once(rrtx_handle* h, const Ctx& c, int nblk, int tpb, bool f32, double* kms, int64_t* launches) {
  {
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (tpb == 64) {
      if (f32)
        hipLaunchKernelGGL(rppk2t::rrt_star_kernel_v2<true>, dim3(nblk), dim3(rppk2t::TPB), 0, h->stream, c, h->v2_chunk_iters);
      else
        hipLaunchKernelGGL(rppk2t::rrt_star_kernel_v2<false>, dim3(nblk), dim3(rppk2t::TPB), 0, h->stream, c, h->v2_chunk_iters);
    } else if (tpb == 128) {
      if (f32)
        hipLaunchKernelGGL(rppk2s::rrt_star_kernel_v2<true>, dim3(nblk), dim3(rppk2s::TPB), 0, h->stream, c, h->v2_chunk_iters);
      else
        hipLaunchKernelGGL(rppk2s::rrt_star_kernel_v2<false>, dim3(nblk), dim3(rppk2s::TPB), 0, h->stream, c, h->v2_chunk_iters);
    } else {
      if (f32)
        hipLaunchKernelGGL(rppk2::rrt_star_kernel_v2<true>, dim3(nblk), dim3(rppk2::TPB), 0, h->stream, c, h->v2_chunk_iters);
      else
        hipLaunchKernelGGL(rppk2::rrt_star_kernel_v2<false>, dim3(nblk), dim3(rppk2::TPB), 0, h->stream, c, h->v2_chunk_iters);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *kms += ms;
    (*launches)++;
  }
  return RRTX_OK;
}
static int launch_rrt_star_v2(rrtx_handle* h, const Ctx& c, int nblk, int tpb, bool f32, double* kms, int64_t* launches) {
  for (int64_t done_it = 0; done_it < c.max_iter; done_it += h->v2_chunk_iters) {
    int rc = launch_rrt_star_v2_once(h, c, nblk, tpb, f32, kms, launches);
    if (rc) return rc;
  }
  return RRTX_OK;
}

// Near-candidate capacity of a shape (rrt_star_v2.hip.h RRT2_NU) and the obstacle tile it holds
static int v2_shape_nu(int tpb) { return tpb == 64 ? rppk2t::NU : tpb == 128 ? rppk2s::NU : rppk2::NU; }
static int v2_shape_maxobs(int tpb) { return tpb == 64 ? rppk2t::MAX_OBS : tpb == 128 ? rppk2s::MAX_OBS : rppk2::MAX_OBS; }

// Expected size of the largest near set of a plan, for a tree that fills the sampling square evenly:
// density * pi * r(n)^2 with r(n) of rrt_04:1329-1334 -> pi * min(ccd^2 ln n, n expand_dis^2) / area, largest at
// n = max_iter + 1; a Poisson tail (6 sigma + 8) on top.  Trees are not even (obstacles, unexplored corners), so this
// only steers the first choice of shape: an instance that still overflows is planned again on the next larger shape.
static int estimate_near_capacity(const rrtx_params& p) {
  const double n = (double)p.max_iter + 1.0;
  const double side = fabs(p.rand_max - p.rand_min);
  const double area = side * side > 1e-12 ? side * side : 1e-12;
  double a = p.connect_circle_dist * p.connect_circle_dist * log(n > 2.0 ? n : 2.0), b = n * p.expand_dis * p.expand_dis;
  const double lam = M_PI * (a < b ? a : b) / area;
  const double cap = lam + 6.0 * sqrt(lam) + 8.0;
  return cap > 1e6 ? 1000000 : (int)cap;
}

extern "C" {

int rrtx_abi_version(void) { return RRTX_ABI_VERSION; }

int rrtx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* rrtx_last_error(rrtx_handle* h) { return h ? h->err.c_str() : "null handle"; }

void rrtx_destroy(rrtx_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->rccl_comm) {
    RcclApi* a = rccl_api();
    if (a->lib) a->CommDestroy((ncclComm_t)h->rccl_comm);
  }
  for (void* q : h->allocs) hipFree(q);
  for (auto& bp : h->big) {
    if (bp.px) hipFree(bp.px);
    if (bp.py) hipFree(bp.py);
    if (bp.pyaw) hipFree(bp.pyaw);
  }
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

static inline bool is_dubins(int algo) { return algo == RRTX_ALGO_DUBINS || algo == RRTX_ALGO_RRT_DUBINS; }
// planners whose nodes are poses with a stored edge polyline (rrt_03 / rrt_05 / rrt_06)
static inline bool is_pose_tree(int algo) { return is_dubins(algo) || algo == RRTX_ALGO_RS; }

int rrtx_create(const rrtx_params* p, rrtx_handle** out) {
  if (!p || !out) return RRTX_E_INVALID;
  *out = nullptr;
  if (p->abi_version != RRTX_ABI_VERSION) return RRTX_E_INVALID;
  if (p->algo != RRTX_ALGO_RRT && p->algo != RRTX_ALGO_RRT_STAR && p->algo != RRTX_ALGO_INFORMED && p->algo != RRTX_ALGO_DUBINS && p->algo != RRTX_ALGO_BITSTAR && p->algo != RRTX_ALGO_RRT_DUBINS && p->algo != RRTX_ALGO_RS) return RRTX_E_INVALID;
  if (p->algo == RRTX_ALGO_RS && (!(p->step_size > 0.0) || !(p->curvature > 0.0))) return RRTX_E_INVALID;
  if (p->n_instances < 1 || p->max_iter < 0 || !(p->path_resolution > 0.0) || !(p->expand_dis >= 0.0))
    return RRTX_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev)
    return RRTX_E_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, p->device) != hipSuccess) return RRTX_E_NO_DEVICE;
  if (!strstr(prop.gcnArchName, "gfx950") && !getenv("RRTX_ALLOW_ANY_ARCH")) return RRTX_E_NO_DEVICE;
  rrtx_handle* h = new rrtx_handle();
  h->p = *p;
  h->device = p->device;
  h->n_inst = p->n_instances;
  memset(&h->stats, 0, sizeof(h->stats));
  memset(&h->c, 0, sizeof(h->c));
  // Iterations per kernel launch.  A launch ends when its slowest instance does, so short chunks leave the chip
  // part idle at the end of every launch: 1024-iteration chunks cost the C2 batch 9 % (103 launches) against one launch,
  // 16384-iteration chunks (7 launches) still 1.9 % (18.47 vs 18.12 s, same box).  A plan of up to 131072 iterations
  // (32768 for the other kernels) is ONE launch now -- C2's 105000 iterations: 18 s of kernel time.
  if (const char* e = getenv("RRTX_CHUNK_ITERS")) {
    h->chunk_iters = atoi(e) > 0 ? atoi(e) : 32768;
    h->v2_chunk_iters = h->chunk_iters;
  }
  *out = h;  // returned even on failure below so the caller can read last_error, then destroy
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreate(&h->ev0));
  HIPCHK(h, hipEventCreate(&h->ev1));
  // node capacity: start + one node per iteration; padded so each wave's 512-node stride stays inside
  // (rrt_06: try_goal_path :1572-1582 can append a second node per iteration)
  const int64_t cap = (p->algo == RRTX_ALGO_RS ? 2 : 1) * (int64_t)p->max_iter + 2;
  h->stride = (cap + 511) / 512 * 512 + 2560;
  const size_t tot = (size_t)h->stride * h->n_inst;
  Ctx& c = h->c;
  int rc;
  if ((rc = dalloc(h, &c.inst, h->n_inst))) return rc;
  if ((rc = dalloc(h, &c.x, tot))) return rc;
  if ((rc = dalloc(h, &c.y, tot))) return rc;
  if ((rc = dalloc(h, &c.cost, tot))) return rc;
  if ((rc = dalloc(h, &c.parent, tot))) return rc;
  if ((rc = dalloc(h, &c.first_child, tot))) return rc;
  if ((rc = dalloc(h, &c.next_sib, tot))) return rc;
  if ((rc = dalloc(h, &c.prev_sib, tot))) return rc;
  if ((rc = dalloc(h, &c.hits, tot))) return rc;
  if ((rc = dalloc(h, &c.stack, tot))) return rc;
  if ((p->algo == RRTX_ALGO_RRT_STAR && p->search_until_max_iter) || p->algo == RRTX_ALGO_INFORMED) {
    // float mirror of the coordinates for the prefiltered streaming passes (rrt_star_v2_body.inc, rrt_informed.hip.h)
    if ((rc = dalloc(h, &c.xf, tot))) return rc;
    if ((rc = dalloc(h, &c.yf, tot))) return rc;
  }
  if (p->algo == RRTX_ALGO_RRT_STAR && p->search_until_max_iter) {
    if ((rc = dalloc(h, &c.elen, tot))) return rc;   // cached parent-edge lengths (cost propagation)
    if ((rc = dalloc(h, &c.xq, tot))) return rc;     // 16-bit mirror (first stage of the streaming pass)
  }
  if (p->algo == RRTX_ALGO_INFORMED)
    if ((rc = dalloc(h, &c.xq, tot))) return rc;     // 16-bit mirror: the one pass per iteration of the rrt_07 kernel
  if ((rc = dalloc(h, &c.results, h->n_inst))) return rc;
  c.path_cap = (int32_t)(cap + 1 < 8192 ? cap + 1 : 8192);
  if ((rc = dalloc(h, &c.path_xy, (size_t)h->n_inst * c.path_cap * 2))) return rc;
  double *dox, *doy, *dothr, *dr2;
  if ((rc = dalloc(h, &dox, rppk::MAX_OBS))) return rc;
  if ((rc = dalloc(h, &doy, rppk::MAX_OBS))) return rc;
  if ((rc = dalloc(h, &dothr, rppk::MAX_OBS))) return rc;
  if ((rc = dalloc(h, &dr2, (size_t)cap + 2))) return rc;
  c.ox = dox;
  c.oy = doy;
  c.othr = dothr;
  c.r2tab = dr2;
  c.m = 0;
  c.stride = h->stride;
  c.algo = p->algo;
  c.sampler = p->sampler;
  c.goal_sample_rate = p->goal_sample_rate;
  c.max_iter = p->max_iter;
  c.has_play = p->has_play_area;
  c.until_max = p->search_until_max_iter;
  c.rand_min = p->rand_min;
  c.rand_max = p->rand_max;
  c.expand_dis = p->expand_dis;
  c.res = p->path_resolution;
  for (int i = 0; i < 4; i++) c.play_area[i] = p->play_area[i];
  c.trace_inst = -1;
  // find_near_nodes radius schedule (rrt_04:1329-1334, :1337): r(nnode)**2 with this host's libm,
  // exactly the expression the reference evaluates per iteration; it depends on nnode only.
  std::vector<double> r2((size_t)cap + 2, 0.0);
  for (int64_t nn = 1; nn < cap + 2; nn++) {
    double r;
    if (p->algo == RRTX_ALGO_INFORMED) {
      r = 50.0 * libm_sqrt(libm_log((double)nn) / (double)nn);   // rrt_07:1139, indexed by len(node_list), no cap
    } else {
      r = p->connect_circle_dist * libm_sqrt(libm_log((double)nn) / (double)nn);
      if (p->expand_dis < r) r = p->expand_dis;
    }
    r2[nn] = py_sq_host(r);
  }
  if (p->algo == RRTX_ALGO_INFORMED) {
    if ((rc = dalloc(h, &h->cbest, h->n_inst))) return rc;
    if ((rc = dalloc(h, &h->d_iargs, h->n_inst))) return rc;
    h->iargs.resize(h->n_inst);
    h->icmin.assign(h->n_inst, p->informed_c_min);
    for (int i = 0; i < h->n_inst; i++) {
      for (int k = 0; k < 4; k++) h->iargs[i].rot[k] = p->informed_rot[k];
      h->iargs[i].c_min2 = py_sq_host(p->informed_c_min);   // c_min ** 2 rrt_07:1147
    }
  }
  memset(&h->ba, 0, sizeof(h->ba));
  if (p->algo == RRTX_ALGO_BITSTAR) {
    rppb::BitArgs& b = h->ba;
    if ((rc = dalloc(h, &b.cfg, h->n_inst))) return rc;
    if ((rc = dalloc(h, &b.dslab, (size_t)rppb::DSLAB * h->n_inst))) return rc;
    if ((rc = dalloc(h, &b.islab, (size_t)rppb::ISLAB * h->n_inst))) return rc;
    if ((rc = dalloc(h, &b.out_i, (size_t)8 * h->n_inst))) return rc;
    if ((rc = dalloc(h, &b.out_g, h->n_inst))) return rc;
    b.trace_inst = -1;
    h->bcfg.resize(h->n_inst);
    for (int i = 0; i < h->n_inst; i++) {
      rpp::BitCfg& c2 = h->bcfg[i];
      memset(&c2, 0, sizeof(c2));
      c2.start[0] = p->start[0]; c2.start[1] = p->start[1];
      c2.goal[0] = p->goal[0]; c2.goal[1] = p->goal[1];
      c2.rand_min = p->rand_min; c2.rand_max = p->rand_max;
      for (int k = 0; k < 4; k++) c2.rot[k] = p->informed_rot[k];
      c2.c_min = p->informed_c_min;
      c2.c_min2 = py_sq_host(p->informed_c_min);
      c2.num_cells = std::ceil((p->rand_max - p->rand_min) / 0.01);   // RTree num_cells (rrt_08:42-44, :165-168)
      c2.max_iter = p->max_iter;
    }
  }
  memset(&h->da, 0, sizeof(h->da));
  if (is_pose_tree(p->algo)) {
    rppd::DubArgs& d = h->da;
    d.plain = p->algo == RRTX_ALGO_RRT_DUBINS;
    // polyline points per instance (edges replaced by rewire stay allocated; rrt_06 edges run all the way to the
    // sample at step_size spacing and are longer)
    int64_t ppn = p->algo == RRTX_ALGO_RS ? 160 : 128;   // polyline points per node, on average
    if (const char* e = getenv("RRTX_POOL_POINTS_PER_NODE"))   // test knob: a small pool exercises the re-plan below
      if (atoi(e) > 0) ppn = atoi(e);
    d.pool_cap = ppn * cap + 8192;
    if ((rc = dalloc(h, &d.yaw, tot))) return rc;
    if ((rc = dalloc(h, &d.poff, tot))) return rc;
    if ((rc = dalloc(h, &d.plen, tot))) return rc;
    if ((rc = dalloc(h, &d.pool_x, (size_t)d.pool_cap * h->n_inst))) return rc;
    if ((rc = dalloc(h, &d.pool_y, (size_t)d.pool_cap * h->n_inst))) return rc;
    if ((rc = dalloc(h, &d.pool_used, h->n_inst))) return rc;
    d.curvature = p->curvature;
    d.goal_yaw_th = p->goal_yaw_th;
    d.goal_xy_th = p->goal_xy_th;
  }
  if (p->algo == RRTX_ALGO_RS) {
    rppd::DubArgs& d = h->da;
    d.step_size = p->step_size;
    if ((rc = dalloc(h, &d.pool_yaw, (size_t)d.pool_cap * h->n_inst))) return rc;
    if ((rc = dalloc(h, &d.dscr, tot))) return rc;
  }
  if (is_dubins(p->algo)) {
    rppd::DubArgs& d = h->da;
    if ((rc = dalloc(h, &d.spx, (size_t)rppd::PMAX * h->n_inst))) return rc;
    if ((rc = dalloc(h, &d.spy, (size_t)rppd::PMAX * h->n_inst))) return rc;
    if ((rc = dalloc(h, &d.plans, (size_t)rppd::NUD * h->n_inst))) return rc;
    if ((rc = dalloc(h, &d.rawslot, tot))) return rc;
  }
  HIPCHK(h, hipMemcpy(dr2, r2.data(), r2.size() * sizeof(double), hipMemcpyHostToDevice));
  // default per-instance state: ctor start/goal, RNG seeded with the instance number
  h->host_inst.resize(h->n_inst);
  for (int i = 0; i < h->n_inst; i++) {
    Inst& I = h->host_inst[i];
    memset(&I, 0, sizeof(I));
    rpp::mt_seed_u64(&I.rng, (uint64_t)i);
    for (int k = 0; k < 3; k++) {
      I.start[k] = p->start[k];
      I.goal[k] = p->goal[k];
    }
  }
  return RRTX_OK;
}

int rrtx_set_obstacles(rrtx_handle* h, const double* oxyr, int32_t m) {
  if (!h || m < 0 || (m > 0 && !oxyr)) return RRTX_E_INVALID;
  if (m > rppk::MAX_OBS) {
    h->err = "more than 256 obstacles";
    return RRTX_E_INVALID;
  }
  if (h->p.algo == RRTX_ALGO_RS && m > rppr::MAX_OBS) {
    h->err = "RRTX_ALGO_RS: more than 64 obstacles";
    return RRTX_E_INVALID;
  }
  std::vector<double> ox(rppk::MAX_OBS, 0.0), oy(rppk::MAX_OBS, 0.0), th(rppk::MAX_OBS, -1.0);
  for (int k = 0; k < m; k++) {
    ox[k] = oxyr[3 * k];
    oy[k] = oxyr[3 * k + 1];
    th[k] = py_sq_host(oxyr[3 * k + 2] + h->p.robot_radius);  // (size+robot_radius)**2  rrt_04:1227
  }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy((void*)h->c.ox, ox.data(), sizeof(double) * rppk::MAX_OBS, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy((void*)h->c.oy, oy.data(), sizeof(double) * rppk::MAX_OBS, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy((void*)h->c.othr, th.data(), sizeof(double) * rppk::MAX_OBS, hipMemcpyHostToDevice));
  h->c.m = m;
  h->m = m;
  h->obst.assign(oxyr, oxyr + 3 * (size_t)m);
  return RRTX_OK;
}

int rrtx_set_rng_state(rrtx_handle* h, int32_t instance, const uint32_t* mt624, int32_t pos) {
  if (!h || !mt624 || instance < 0 || instance >= h->n_inst || pos < 0 || pos > 624) return RRTX_E_INVALID;
  h->inst_dirty = true;
  memcpy(h->host_inst[instance].rng.mt, mt624, 624 * 4);
  h->host_inst[instance].rng.pos = pos;
  return RRTX_OK;
}

int rrtx_get_rng_state(rrtx_handle* h, int32_t instance, uint32_t* mt624, int32_t* pos) {
  if (!h || !mt624 || !pos || instance < 0 || instance >= h->n_inst) return RRTX_E_INVALID;
  if (h->planned) {
    HIPCHK(h, hipSetDevice(h->device));
    rpp::MT r;
    HIPCHK(h, hipMemcpy(&r, &h->c.inst[instance].rng, sizeof(r), hipMemcpyDeviceToHost));
    memcpy(mt624, r.mt, 624 * 4);
    *pos = r.pos;
  } else {
    memcpy(mt624, h->host_inst[instance].rng.mt, 624 * 4);
    *pos = h->host_inst[instance].rng.pos;
  }
  return RRTX_OK;
}

int rrtx_seed_instances(rrtx_handle* h, int32_t first, int32_t count, const uint64_t* seeds) {
  if (!h || !seeds || first < 0 || count < 0 || first + count > h->n_inst) return RRTX_E_INVALID;
  h->inst_dirty = true;
  for (int i = 0; i < count; i++) rpp::mt_seed_u64(&h->host_inst[first + i].rng, seeds[i]);
  return RRTX_OK;
}

int rrtx_set_instance(rrtx_handle* h, int32_t instance, const double* start3, const double* goal3) {
  if (!h || instance < 0 || instance >= h->n_inst) return RRTX_E_INVALID;
  h->inst_dirty = true;
  Inst& I = h->host_inst[instance];
  if (start3) {
    I.start[0] = start3[0];
    I.start[1] = start3[1];
    if (is_pose_tree(h->p.algo)) I.start[2] = start3[2];   // yaw (rrt_05:1406, rrt_06:1518)
    if (h->p.algo == RRTX_ALGO_BITSTAR) {
      h->bcfg[instance].start[0] = start3[0];
      h->bcfg[instance].start[1] = start3[1];
    }
  }
  if (goal3) {
    I.goal[0] = goal3[0];
    I.goal[1] = goal3[1];
    if (is_pose_tree(h->p.algo)) I.goal[2] = goal3[2];
    if (h->p.algo == RRTX_ALGO_BITSTAR) {
      h->bcfg[instance].goal[0] = goal3[0];
      h->bcfg[instance].goal[1] = goal3[1];
    }
  }
  return RRTX_OK;
}

int rrtx_set_instance_rotation(rrtx_handle* h, int32_t instance, const double* rot4, double c_min) {
  if (!h || !rot4 || instance < 0 || instance >= h->n_inst) return RRTX_E_INVALID;
  if (h->p.algo == RRTX_ALGO_INFORMED) {
    for (int k = 0; k < 4; k++) h->iargs[instance].rot[k] = rot4[k];
    h->iargs[instance].c_min2 = py_sq_host(c_min);   // c_min ** 2 rrt_07:1147
    h->icmin[instance] = c_min;
    return RRTX_OK;
  }
  if (h->p.algo != RRTX_ALGO_BITSTAR) return RRTX_E_STATE;
  rpp::BitCfg& c2 = h->bcfg[instance];
  for (int k = 0; k < 4; k++) c2.rot[k] = rot4[k];
  c2.c_min = c_min;
  c2.c_min2 = py_sq_host(c_min);
  return RRTX_OK;
}

int rrtx_enable_trace(rrtx_handle* h, int32_t instance) {
  if (!h || instance < -1 || instance >= h->n_inst) return RRTX_E_INVALID;
  h->trace_inst = instance;
  if (instance >= 0 && !h->c.tr_rx) {
    int rc;
    const size_t n = (size_t)h->p.max_iter + 1;
    if ((rc = dalloc(h, &h->c.tr_rx, n))) return rc;
    if ((rc = dalloc(h, &h->c.tr_ry, n))) return rc;
    if ((rc = dalloc(h, &h->c.tr_near, n))) return rc;
    if ((rc = dalloc(h, &h->c.tr_nn, n))) return rc;
    if ((rc = dalloc(h, &h->c.tr_kind, n))) return rc;
    HIPCHK(h, hipMemset(h->c.tr_kind, 0, sizeof(int32_t) * n));
  }
  h->c.trace_inst = instance;
  return RRTX_OK;
}

static int plan_finish(rrtx_handle* h);

int rrtx_plan_begin(rrtx_handle* h) {
  if (!h) return RRTX_E_INVALID;
  h->run = rrtx_handle::Run();
  rrtx_handle::Run& R = h->run;
  double &kms = R.kms, &kms_main = R.kms_main;
  int64_t &launches = R.launches, &launches_main = R.launches_main;
  std::vector<Result>& res = R.res;
  bool &use_v2 = R.use_v2, &v2_f32 = R.v2_f32;
  int& v2_tpb = R.v2_tpb;
  (void)kms; (void)kms_main; (void)launches; (void)launches_main; (void)res; (void)use_v2; (void)v2_f32; (void)v2_tpb;
  R.t0 = std::chrono::steady_clock::now();
  h->planned = false;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());   // uploads made through the null stream (obstacles, tables) are complete
  Ctx& c = h->c;
  const int B = h->n_inst;
  h->stats_retried = 0;
  // f32-mirror margin = 2^-20 * the largest coordinate magnitude a node or sample is assumed to have (see scan2f);
  // rrt_07's informed samples are not clipped to the sampling square, so twice that (the kernel checks and falls back)
  {
    double mag = fabs(c.rand_min) > fabs(c.rand_max) ? fabs(c.rand_min) : fabs(c.rand_max);
    for (int i = 0; i < B; i++) {
      const Inst& I = h->host_inst[i];
      const double v[4] = {I.start[0], I.start[1], I.goal[0], I.goal[1]};
      for (double q : v)
        if (fabs(q) > mag) mag = fabs(q);
    }
    if (c.algo == RRTX_ALGO_INFORMED) mag *= 2.0;
    c.f32_m = ldexp(mag > 1.0 ? mag : 1.0, -20);
    // 16-bit mirror: the square [lo, hi]^2 that holds every node and sample (sampling square, starts, goals; nodes are
    // convex combinations of those).  Quantisation error <= step/2 per coordinate -> a point moves by <= step/sqrt(2);
    // node and query are both on the grid -> distance error < sqrt(2) steps; the arithmetic is exact (integer).
    double lo = c.rand_min < c.rand_max ? c.rand_min : c.rand_max, hi = c.rand_min < c.rand_max ? c.rand_max : c.rand_min;
    for (int i = 0; i < B; i++) {
      const Inst& I = h->host_inst[i];
      const double v[4] = {I.start[0], I.start[1], I.goal[0], I.goal[1]};
      for (double q : v) {
        if (q < lo) lo = q;
        if (q > hi) hi = q;
      }
    }
    if (c.algo == RRTX_ALGO_INFORMED) {
      // rrt_07's informed samples are not clipped to the sampling square (the ellipse of :1145-1159 may reach past it) and
      // a node may step expand_dis beyond its sample: an eighth of the range on every side.  A node that still leaves the
      // grid switches its instance to the f32 / f64 passes (rrt_informed.hip.h)
      double frac = 0.125;
      if (const char* e = getenv("RRTX_Q16_PAD")) frac = atof(e);   // test knob: a negative margin makes nodes leave the grid
      const double pad = frac * (hi - lo > 1e-9 ? hi - lo : 1.0) + (frac >= 0.0 ? fabs(c.expand_dis) : 0.0);
      lo -= pad;
      hi += pad;
    }
    const double range = hi - lo > 1e-9 ? hi - lo : 1.0;
    c.q_lo = lo;
    c.q_step = range / 65535.0;
    c.q_inv = 65535.0 / range;
    c.q_m = 1.4375 * c.q_step;   // node AND query rounded to the grid: sqrt(2) steps (scan2q)
  }
  if (const char* e = getenv("RRTX_F32"))
    if (atoi(e) == 0 && c.algo == RRTX_ALGO_INFORMED) c.xf = c.yf = nullptr;   // informed kernel: f64 passes only
  if (const char* e = getenv("RRTX_Q16"))
    if (atoi(e) == 0) c.xq = nullptr;   // rrt_04 kernel: no 16-bit first stage (f32 mirror first)
  // rrt_04 kernel, one-wave shape: a streaming pass serves up to 1 + spec2 iterations (clamped to the kernel's RRT2_SPECK;
  // RRTX_SPEC2=0: one pass per iteration)
  c.spec2 = 8;
  if (const char* e = getenv("RRTX_SPEC2")) c.spec2 = atoi(e) > 0 ? atoi(e) : 0;
  // The staged per-instance start state (RNG, start / goal) lives on the device too: uploaded when the host changed it,
  // copied device -> device at every plan (2.7 KB per instance: 44 MB of pageable-memory upload per plan of 16 384 instances)
  if (!h->d_inst0) {
    int rc2 = dalloc(h, &h->d_inst0, B);
    if (rc2) return rc2;
    h->inst_dirty = true;
  }
  if (h->inst_dirty) {
    HIPCHK(h, hipMemcpyAsync(h->d_inst0, h->host_inst.data(), sizeof(Inst) * B, hipMemcpyHostToDevice, h->stream));
    h->inst_dirty = false;
  }
  HIPCHK(h, hipMemcpyAsync(c.inst, h->d_inst0, sizeof(Inst) * B, hipMemcpyDeviceToDevice, h->stream));
  {
    dim3 g(64, B);
    hipLaunchKernelGGL(rppk::rrt_init_kernel, g, dim3(256), 0, h->stream, c);
    hipLaunchKernelGGL(rppk::rrt_root_kernel, dim3((B + 63) / 64), dim3(64), 0, h->stream, c, B);
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  res.assign(B, Result());
  if (c.algo == RRTX_ALGO_BITSTAR) {
    // BIT*: one launch, one lane per instance (rrt_bitstar.hip.h); obstacle thresholds are size ** 2 (rrt_08:381)
    for (int i = 0; i < B; i++) {
      h->bcfg[i].m = c.m;
      h->bcfg[i].ox = c.ox;
      h->bcfg[i].oy = c.oy;
      h->bcfg[i].othr = c.othr;
    }
    HIPCHK(h, hipMemcpyAsync(h->ba.cfg, h->bcfg.data(), sizeof(rpp::BitCfg) * B, hipMemcpyHostToDevice, h->stream));
    if (h->trace_inst >= 0 && !h->ba.tr_a) {
      int rc2;
      if ((rc2 = dalloc(h, &h->ba.tr_a, 1 << 16))) return rc2;
      if ((rc2 = dalloc(h, &h->ba.tr_b, 1 << 16))) return rc2;
      h->ba.tr_cap = 1 << 16;
    }
    h->ba.trace_inst = h->trace_inst;
    // one wave per instance when the per-vertex state fits LDS (rrt_bitstar_wave.hip.h); else one lane per instance
    const char* bk = getenv("RRTX_BITSTAR");
    R.bit_wave = c.m <= rppb::OB && c.max_iter + 2 <= rppb::VL && !(bk && !strcmp(bk, "lane"));
    if (!h->bit_queue) {
      int rc2;
      if ((rc2 = dalloc(h, &h->bit_queue, B))) return rc2;
      if ((rc2 = dalloc(h, &h->bit_qhead, 1))) return rc2;
      if ((rc2 = dalloc(h, &h->ba.save_i, (size_t)8 * B))) return rc2;
    }
    if (const char* e = getenv("RRTX_BITSTAR_TRIPS")) h->bit_trip_bound = atoi(e) > 0 ? atoi(e) : 20000;
    R.pending.resize(B);
    for (int i = 0; i < B; i++) R.pending[i] = i;
    // Queue order = longest expected run first: BIT* run times have a heavy tail (p99 = 8 x the mean) and grow as start and
    // goal get closer (a small informed set is resampled densely: correlation -0.36 with the distance over 400 instances of
    // the C4 generator), so the instances most likely to be the last ones running start first.  Results do not depend on it.
    if (!getenv("RRTX_BITSTAR_FIFO"))
      std::stable_sort(R.pending.begin(), R.pending.end(),
                       [&](int32_t a, int32_t b) { return h->bcfg[a].c_min < h->bcfg[b].c_min; });
  }
  // RRT* with search_until_max_iter: the latency-lean iteration kernel runs every iteration; the general kernel
  // below then only performs the final goal search (rrt_04:1080-1084).  RRTX_KERNEL=v1 forces the general kernel.
  const char* kv = getenv("RRTX_KERNEL");
  use_v2 = c.algo == RRTX_ALGO_RRT_STAR && c.until_max && !(kv && !strcmp(kv, "v1"));
  if (use_v2) {
    // workgroup shape: fewer threads per instance once more instances want to be resident (8 / 16 workgroups per CU),
    // as long as the shape's obstacle tile and near-candidate capacity fit the problem
    const int need_nu = estimate_near_capacity(h->p);
    int tpb = 256;
    if (B > 1280 && c.m <= rppk2s::MAX_OBS && need_nu <= rppk2s::NU) tpb = 128;
    if (B > 2560 && c.m <= rppk2t::MAX_OBS && need_nu <= rppk2t::NU) tpb = 64;
    if (const char* e = getenv("RRTX_TPB")) {
      const int v = atoi(e);
      tpb = (v == 64 && c.m <= rppk2t::MAX_OBS) ? 64 : (v == 128 && c.m <= rppk2s::MAX_OBS) ? 128 : 256;
    }
    // f32-mirror prefilter (default on; RRTX_F32=0 streams the f64 arrays): margin = 2^-20 * largest coordinate
    // magnitude any node or sample can have (see scan2f)
    bool f32 = c.xf != nullptr;
    if (const char* e = getenv("RRTX_F32")) f32 = f32 && atoi(e) != 0;
    v2_tpb = tpb;
    v2_f32 = f32;
  }
  if (c.algo == RRTX_ALGO_INFORMED) {
    std::vector<double> inf(B, INFINITY);
    HIPCHK(h, hipMemcpyAsync(h->cbest, inf.data(), sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));   // `inf` is a local
  }
  // on the handle's own stream: it is a non-blocking stream, work queued on the null stream is not ordered with it
  if (is_pose_tree(c.algo)) {
    HIPCHK(h, hipMemsetAsync(h->da.pool_used, 0, sizeof(int64_t) * B, h->stream));
    h->pool_loc.assign(B, rrtx_handle::PoolLoc());
    for (int i = 0; i < B; i++) {
      rrtx_handle::PoolLoc& pl = h->pool_loc[i];
      pl.px = h->da.pool_x; pl.py = h->da.pool_y; pl.pyaw = h->da.pool_yaw;
      pl.cap = h->da.pool_cap; pl.slab = i;
    }
    h->da.pool_slot = nullptr;
  }
  if (const char* e = getenv("RRTX_RS_EAGER")) h->da.eager = atoi(e) != 0;
  if (const char* e = getenv("RRTX_INFORMED_EAGER")) h->informed_eager = atoi(e) != 0;   // rrt_07: test every near candidate like the reference
  if (const char* e = getenv("RRTX_INFORMED_EXACT_SEG"))   // rrt_07 test knob: no tolerance bands -- every verdict from the exact segment form, every candidate list from the exact **2 form
    h->informed_eager = (h->informed_eager & 1) | (atoi(e) != 0 ? 6 : 0);
  h->da.lazy = 0;
  h->da.filter = 1;
  if (const char* e = getenv("RRTX_DUBINS_FILTER")) h->da.filter = atoi(e) != 0;
  if (const char* e = getenv("RRTX_DUBINS_LAZY")) h->da.lazy = atoi(e) != 0;
  if (c.algo == RRTX_ALGO_INFORMED) {
    for (int i = 0; i < B; i++) {
      const Inst& I = h->host_inst[i];
      h->iargs[i].xc[0] = (I.start[0] + I.goal[0]) / 2.0;   // x_center rrt_07:1056-1057
      h->iargs[i].xc[1] = (I.start[1] + I.goal[1]) / 2.0;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_iargs, h->iargs.data(), sizeof(rppi::InformedArgs) * B, hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  R.stage = use_v2 ? 1 : 2;
  return RRTX_OK;
}

int rrtx_plan_step(rrtx_handle* h, int32_t* n_pending) {
  if (!h) return RRTX_E_INVALID;
  if (h->run.stage == 0) return RRTX_E_STATE;
  rrtx_handle::Run& R = h->run;
  double &kms = R.kms, &kms_main = R.kms_main;
  int64_t &launches = R.launches, &launches_main = R.launches_main;
  std::vector<Result>& res = R.res;
  bool &use_v2 = R.use_v2, &v2_f32 = R.v2_f32;
  int& v2_tpb = R.v2_tpb;
  (void)kms; (void)kms_main; (void)launches; (void)launches_main; (void)res; (void)use_v2; (void)v2_f32; (void)v2_tpb;
  HIPCHK(h, hipSetDevice(h->device));
  Ctx& c = h->c;
  const int B = h->n_inst;
  R.steps++;
  if (n_pending) *n_pending = B;
  if (R.stage == 1) {
    // RRT* (rrt_04, search_until_max_iter): one launch of the iteration kernel = v2_chunk_iters iterations of every instance
    int rc2 = launch_rrt_star_v2_once(h, c, B, v2_tpb, v2_f32, &kms, &launches);
    if (rc2) return rc2;
    R.v2_done_it += h->v2_chunk_iters;
    if (R.v2_done_it >= c.max_iter) {
      kms_main = kms;
      launches_main = launches;
      R.stage = 2;   // the general kernel then performs the final goal search (rrt_04:1080-1084)
    }
    return RRTX_OK;
  }
  if (c.algo == RRTX_ALGO_BITSTAR) {
    // One BOUNDED launch over the pending instances: persistent waves pull them from the device-side queue; an instance
    // that uses up its trips is carried over (its state stays in its slab) and queued again for the next launch
    const int np = (int)R.pending.size();
    const int32_t zero = 0;
    HIPCHK(h, hipMemcpyAsync(h->bit_queue, R.pending.data(), sizeof(int32_t) * np, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->bit_qhead, &zero, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    rppb::BitArgs ba = h->ba;
    ba.queue = h->bit_queue;
    ba.qhead = h->bit_qhead;
    ba.n_pending = np;
    ba.trip_bound = h->bit_trip_bound;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (R.bit_wave) {
      // as many waves as the chip keeps resident (LDS: 9 workgroups per CU), never more than there are instances
      hipDeviceProp_t prop;
      HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
      int grid = prop.multiProcessorCount * 9;
      if (const char* e = getenv("RRTX_BITSTAR_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;
      if (grid > np) grid = np;
      hipLaunchKernelGGL(rppb::bitstar_wave_kernel, dim3(grid), dim3(64), 0, h->stream, ba, c.inst, c.results, B);
    } else {
      hipLaunchKernelGGL(rppb::bitstar_kernel, dim3((B + 63) / 64), dim3(64), 0, h->stream, ba, c.inst, c.results, B);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipMemcpyAsync(res.data(), c.results, sizeof(Result) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    kms += ms;
    launches++;
    std::vector<int32_t> left;
    for (int32_t i : R.pending)
      if (!(res[i].status & RRTX_ST_DONE)) left.push_back(i);
    R.pending.swap(left);
    if (n_pending) *n_pending = (int32_t)R.pending.size();
    if (!R.pending.empty()) {
      if (launches > 4000000LL / h->bit_trip_bound + 16) {
        h->err = "BIT* kernel did not converge to DONE";
        return RRTX_E_STATE;
      }
      return RRTX_OK;
    }
    if (n_pending) *n_pending = 0;
    return plan_finish(h);
  }
  {
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    if (c.algo == RRTX_ALGO_INFORMED)
      hipLaunchKernelGGL((rppi::rrt_informed_kernel<rppi::NUI_SMALL, 4>), dim3(B), dim3(rppi::TPB), 0, h->stream, c,
                         h->d_iargs, h->cbest, h->chunk_iters, h->informed_eager);
    else if (is_dubins(c.algo))
      hipLaunchKernelGGL(rppd::rrt_dubins_kernel, dim3(B), dim3(rppd::TPB), 0, h->stream, c, h->da, h->chunk_iters);
    else if (c.algo == RRTX_ALGO_RS)
      hipLaunchKernelGGL(rppr::rrt_rs_kernel, dim3(B), dim3(rppr::TPB), 0, h->stream, c, h->da, h->chunk_iters);
    else
      hipLaunchKernelGGL(rppk::rrt_plan_kernel, dim3(B), dim3(rppk::TPB), 0, h->stream, c, h->chunk_iters);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipMemcpyAsync(res.data(), c.results, sizeof(Result) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    kms += ms;
    launches++;
    int left = 0;
    for (int i = 0; i < B; i++)
      if (!(res[i].status & RRTX_ST_DONE)) left++;
    if (n_pending) *n_pending = left;
    if (left) {
      if (launches > (int64_t)h->p.max_iter / h->chunk_iters + 8 + launches_main) {
        h->err = "planner kernel did not converge to DONE";
        return RRTX_E_STATE;
      }
      return RRTX_OK;
    }
  }
  return plan_finish(h);
}

// Everything after the last instance has finished its main kernel: re-plans of instances that outgrew a fixed table, the
// counters, the return code
static int plan_finish(rrtx_handle* h) {
  rrtx_handle::Run& R = h->run;
  double &kms = R.kms, &kms_main = R.kms_main;
  int64_t &launches = R.launches, &launches_main = R.launches_main;
  std::vector<Result>& res = R.res;
  bool &use_v2 = R.use_v2, &v2_f32 = R.v2_f32;
  int& v2_tpb = R.v2_tpb;
  (void)kms; (void)kms_main; (void)launches; (void)launches_main; (void)res; (void)use_v2; (void)v2_f32; (void)v2_tpb;
  Ctx& c = h->c;
  const int B = h->n_inst;
  R.stage = 0;
  // RRT* iteration kernel: an instance whose near set outgrew the LDS candidate table of its workgroup shape
  // (RRTX_ST_OVERFLOW) is planned again, from its staged start state, on the next larger shape (44 -> 128 -> 256
  // candidates), and finally by the general kernel (512).  Same results as a first plan on that shape: every shape
  // runs the same statements.
  if (use_v2 && !getenv("RRTX_NO_RETRY")) {
    // plans the instances `redo` again: shape = workgroup shape of the iteration kernel, 0 = general kernel alone
    auto replan = [&](const std::vector<int32_t>& redo, int shape) -> int {
      const int nr = (int)redo.size();
      if (!h->inst_map) {
        int rc2;
        if ((rc2 = dalloc(h, &h->inst_map, B))) return rc2;
      }
      HIPCHK(h, hipMemcpyAsync(h->inst_map, redo.data(), sizeof(int32_t) * nr, hipMemcpyHostToDevice, h->stream));
      for (int k = 0; k < nr; k++)
        HIPCHK(h, hipMemcpyAsync(c.inst + redo[k], &h->host_inst[redo[k]], sizeof(Inst), hipMemcpyHostToDevice, h->stream));
      Ctx cr = c;
      cr.inst_map = h->inst_map;
      hipLaunchKernelGGL(rppk::rrt_init_kernel, dim3(64, nr), dim3(256), 0, h->stream, cr);
      hipLaunchKernelGGL(rppk::rrt_root_kernel, dim3((nr + 63) / 64), dim3(64), 0, h->stream, cr, nr);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipStreamSynchronize(h->stream));   // `redo` is read by the copies above
      if (shape) {
        int rc2 = launch_rrt_star_v2(h, cr, nr, shape, v2_f32, &kms, &launches);
        if (rc2) return rc2;
      }
      for (int64_t guard = 0;; guard++) {
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        hipLaunchKernelGGL(rppk::rrt_plan_kernel, dim3(nr), dim3(rppk::TPB), 0, h->stream, cr, h->chunk_iters);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipMemcpyAsync(res.data(), c.results, sizeof(Result) * B, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        kms += ms;
        launches++;
        bool all = true;
        for (int k = 0; k < nr; k++)
          if (!(res[redo[k]].status & RRTX_ST_DONE)) all = false;
        if (all) break;
        if (guard > (int64_t)h->p.max_iter / h->chunk_iters + 8) {
          h->err = "planner kernel did not converge to DONE (retry)";
          return RRTX_E_STATE;
        }
      }
      h->stats_retried += nr;
      return RRTX_OK;
    };
    int shape = v2_tpb;   // 0 = general kernel
    for (;;) {
      std::vector<int32_t> redo;
      for (int i = 0; i < B; i++)
        if (res[i].status & RRTX_ST_OVERFLOW) redo.push_back(i);
      if (redo.empty() || shape == 0) break;
      shape = shape == 64 ? 128 : shape == 128 ? 256 : 0;
      if (shape && c.m > v2_shape_maxobs(shape)) continue;
      int rc2 = replan(redo, shape);
      if (rc2) return rc2;
    }
    // rewire moved a node while near_inds had repeated entries (rrt_04:1337 with :1372): the iteration kernel does not
    // walk the raw list (RRTX_ST_UNSUPPORTED in its status word), the general kernel does (rppk::rewire_raw_walk)
    std::vector<int32_t> redo;
    for (int i = 0; i < B; i++)
      if (res[i].status & RRTX_ST_UNSUPPORTED) redo.push_back(i);
    if (!redo.empty()) {
      int rc2 = replan(redo, 0);
      if (rc2) return rc2;
    }
  }
  // Pose planners (rrt_03 / rrt_05 / rrt_06): an instance that ran out of polyline pool (edges replaced by rewire stay
  // allocated) is planned again, from its staged start state, with a pool four times as large -- twice if need be.
  // Overflows of the other fixed tables (near candidates, points per edge) are not helped by that and stay reported.
  if (is_pose_tree(c.algo) && !getenv("RRTX_NO_RETRY")) {
    int64_t big_cap = h->da.pool_cap;
    for (int attempt = 0; attempt < 2; attempt++) {
      std::vector<int32_t> redo;
      for (int i = 0; i < B; i++)
        if (res[i].status & RRTX_ST_OVERFLOW) redo.push_back(i);
      if (redo.empty()) break;
      const int nr = (int)redo.size();
      big_cap *= 4;
      rrtx_handle::BigPool& bp = h->big[attempt];
      if (bp.cap != big_cap || bp.slabs < nr) {
        // (re)allocate this level: nothing of the CURRENT plan lives in it yet (its users are decided below), and the
        // previous plan's polylines are gone with the re-initialisation above
        if (bp.px) hipFree(bp.px);
        if (bp.py) hipFree(bp.py);
        if (bp.pyaw) hipFree(bp.pyaw);
        bp = rrtx_handle::BigPool();
        const size_t bytes = sizeof(double) * (size_t)big_cap * nr;
        if (hipMalloc((void**)&bp.px, bytes) != hipSuccess || hipMalloc((void**)&bp.py, bytes) != hipSuccess ||
            (c.algo == RRTX_ALGO_RS && hipMalloc((void**)&bp.pyaw, bytes) != hipSuccess)) {
          (void)hipGetLastError();   // no room for the larger pool: the instances keep their RRTX_ST_OVERFLOW
          if (bp.px) hipFree(bp.px);
          if (bp.py) hipFree(bp.py);
          if (bp.pyaw) hipFree(bp.pyaw);
          bp = rrtx_handle::BigPool();
          break;
        }
        bp.cap = big_cap;
        bp.slabs = nr;
      }
      double *bx = bp.px, *by = bp.py, *bw = bp.pyaw;
      int rc2;
      if (!h->inst_map && (rc2 = dalloc(h, &h->inst_map, B))) return rc2;
      if (!h->pool_slot && (rc2 = dalloc(h, &h->pool_slot, B))) return rc2;
      std::vector<int32_t> slot(B, 0);
      for (int k = 0; k < nr; k++) slot[redo[k]] = k;
      HIPCHK(h, hipMemcpyAsync(h->inst_map, redo.data(), sizeof(int32_t) * nr, hipMemcpyHostToDevice, h->stream));
      HIPCHK(h, hipMemcpyAsync(h->pool_slot, slot.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, h->stream));
      for (int k = 0; k < nr; k++) {
        HIPCHK(h, hipMemcpyAsync(c.inst + redo[k], &h->host_inst[redo[k]], sizeof(Inst), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->da.pool_used + redo[k], 0, sizeof(int64_t), h->stream));
      }
      Ctx cr = c;
      cr.inst_map = h->inst_map;
      rppd::DubArgs dr = h->da;
      dr.pool_x = bx; dr.pool_y = by; dr.pool_yaw = bw;
      dr.pool_cap = big_cap;
      dr.pool_slot = h->pool_slot;
      hipLaunchKernelGGL(rppk::rrt_init_kernel, dim3(64, nr), dim3(256), 0, h->stream, cr);
      hipLaunchKernelGGL(rppk::rrt_root_kernel, dim3((nr + 63) / 64), dim3(64), 0, h->stream, cr, nr);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipStreamSynchronize(h->stream));
      for (int64_t guard = 0;; guard++) {
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        if (c.algo == RRTX_ALGO_RS)
          hipLaunchKernelGGL(rppr::rrt_rs_kernel, dim3(nr), dim3(rppr::TPB), 0, h->stream, cr, dr, h->chunk_iters);
        else
          hipLaunchKernelGGL(rppd::rrt_dubins_kernel, dim3(nr), dim3(rppd::TPB), 0, h->stream, cr, dr, h->chunk_iters);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipMemcpyAsync(res.data(), c.results, sizeof(Result) * B, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        kms += ms;
        launches++;
        bool all = true;
        for (int k = 0; k < nr; k++)
          if (!(res[redo[k]].status & RRTX_ST_DONE)) all = false;
        if (all) break;
        if (guard > (int64_t)h->p.max_iter / h->chunk_iters + 8) {
          h->err = "planner kernel did not converge to DONE (pool retry)";
          return RRTX_E_STATE;
        }
      }
      for (int k = 0; k < nr; k++) {
        rrtx_handle::PoolLoc& pl = h->pool_loc[redo[k]];
        pl.px = bx; pl.py = by; pl.pyaw = bw;
        pl.cap = big_cap; pl.slab = k;
      }
      h->stats_retried += nr;
    }
  }
  // Informed RRT*: the near radius of rrt_07:1139 is not capped, so a near set can outgrow the 512 LDS candidate slots
  // of the product shape; those instances are planned again, from their staged start state, on the 2048-slot shape
  // (one workgroup per CU).  Same statements, same results as a first plan on that shape.
  if (c.algo == RRTX_ALGO_INFORMED && !getenv("RRTX_NO_RETRY")) {
    std::vector<int32_t> redo;
    for (int i = 0; i < B; i++)
      if (res[i].status & RRTX_ST_OVERFLOW) redo.push_back(i);
    if (!redo.empty()) {
      const int nr = (int)redo.size();
      if (!h->inst_map) {
        int rc2;
        if ((rc2 = dalloc(h, &h->inst_map, B))) return rc2;
      }
      HIPCHK(h, hipMemcpyAsync(h->inst_map, redo.data(), sizeof(int32_t) * nr, hipMemcpyHostToDevice, h->stream));
      const double inf1 = INFINITY;
      for (int k = 0; k < nr; k++) {
        HIPCHK(h, hipMemcpyAsync(c.inst + redo[k], &h->host_inst[redo[k]], sizeof(Inst), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->cbest + redo[k], &inf1, sizeof(double), hipMemcpyHostToDevice, h->stream));
      }
      Ctx cr = c;
      cr.inst_map = h->inst_map;
      hipLaunchKernelGGL(rppk::rrt_init_kernel, dim3(64, nr), dim3(256), 0, h->stream, cr);
      hipLaunchKernelGGL(rppk::rrt_root_kernel, dim3((nr + 63) / 64), dim3(64), 0, h->stream, cr, nr);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipStreamSynchronize(h->stream));
      for (int64_t guard = 0;; guard++) {
        HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        hipLaunchKernelGGL((rppi::rrt_informed_kernel<rppi::NUI_LARGE, 1>), dim3(nr), dim3(rppi::TPB), 0, h->stream, cr,
                           h->d_iargs, h->cbest, h->chunk_iters, h->informed_eager);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipMemcpyAsync(res.data(), c.results, sizeof(Result) * B, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        kms += ms;
        launches++;
        bool all = true;
        for (int k = 0; k < nr; k++)
          if (!(res[redo[k]].status & RRTX_ST_DONE)) all = false;
        if (all) break;
        if (guard > (int64_t)h->p.max_iter / h->chunk_iters + 8) {
          h->err = "planner kernel did not converge to DONE (overflow retry)";
          return RRTX_E_STATE;
        }
      }
      h->stats_retried += nr;
    }
  }
  // aggregate counters
  // summed on the device (rppk::stats_reduce_kernel): one small record comes back instead of every Inst
  if (!h->d_acc) {
    int rc2 = dalloc(h, &h->d_acc, 1);
    if (rc2) return rc2;
  }
  rppk::StatsAcc acc;
  HIPCHK(h, hipMemsetAsync(h->d_acc, 0, sizeof(rppk::StatsAcc), h->stream));
  hipLaunchKernelGGL(rppk::stats_reduce_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, c.inst, B, h->d_acc);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(&acc, h->d_acc, sizeof(acc), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  rrtx_stats& s = h->stats;
  memset(&s, 0, sizeof(s));
  s.iterations = acc.sum[0];
  s.edges_unique = acc.sum[1];
  s.edges_ref = acc.sum[2];
  s.near_hits = acc.sum[3];
  s.near_unique = acc.sum[4];
  s.rewires = acc.sum[5];
  s.propagated = acc.sum[6];
  s.scan_nodes = acc.sum[7];
  s.algorithmic_bytes = acc.sum[8];
  s.exact_rescans = acc.sum[9];
  s.algorithmic_bytes_two_scan = acc.sum[10];
  s.total_nodes = acc.sum[11];
  s.f32_fallbacks = acc.sum[12];
  s.q16_fallbacks = acc.sum[13];
  s.passes_shared = acc.sum[14];
  s.near_unique_max = acc.nu_max;
  for (int k = 0; k < 16; k++) h->phase[k] = acc.phase[k];
  const bool overflow = (acc.status_or & RRTX_ST_OVERFLOW) != 0, unsupported = (acc.status_or & RRTX_ST_UNSUPPORTED) != 0,
             raises = (acc.status_or & RRTX_ST_REF_RAISES) != 0;
  s.launches = launches;
  s.kernel_ms = kms;
  s.launches_main = kms_main >= 0.0 ? launches_main : launches;
  s.kernel_ms_main = kms_main >= 0.0 ? kms_main : kms;
  s.replanned = h->stats_retried;
  s.main_shape = use_v2 ? v2_tpb
                        : c.algo == RRTX_ALGO_INFORMED ? rppi::TPB
                        : is_dubins(c.algo)            ? rppd::TPB
                        : c.algo == RRTX_ALGO_RS       ? rppr::TPB
                        : c.algo == RRTX_ALGO_BITSTAR  ? 64
                                                       : rppk::TPB;
  s.main_f32 = use_v2 ? (v2_f32 ? 1 : 0) : 0;
  s.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - R.t0).count();
  h->planned = true;
  // Per-instance conditions are per-instance results: the status word of each instance carries them
  // (rrtx_get_results), the other instances' trees are complete and valid.
  if (overflow || raises || unsupported) {
    h->err.clear();
    if (overflow)
      h->err += "RRTX_ST_OVERFLOW: a fixed on-device capacity was exceeded (near-candidate list, polyline pool, or a "
                "BIT* slab); ";
    if (raises)
      h->err += "RRTX_ST_REF_RAISES: the reference raises inside reeds_shepp_path_planning (ZeroDivisionError "
                "rrt_06:1183/:1207 or a math domain error); ";
    if (unsupported) h->err += "RRTX_ST_UNSUPPORTED: a reference code path the kernel does not restate was reached; ";
    h->err += "the affected instances carry the bit in their status word and have no result, all others are complete";
    return RRTX_PARTIAL;
  }
  return RRTX_OK;
}

int rrtx_plan(rrtx_handle* h) {
  int rc = rrtx_plan_begin(h);
  if (rc < 0) return rc;
  int32_t pending = 1;
  while (h->run.stage != 0) {
    rc = rrtx_plan_step(h, &pending);
    if (rc < 0) {
      h->run.stage = 0;
      return rc;
    }
  }
  return rc;
}

int rrtx_set_launch_bound(rrtx_handle* h, int32_t iterations) {
  if (!h || iterations < 1) return RRTX_E_INVALID;
  if (h->run.stage != 0) return RRTX_E_STATE;
  h->chunk_iters = iterations;
  h->v2_chunk_iters = iterations;
  h->bit_trip_bound = iterations;
  return RRTX_OK;
}

int rrtx_get_tree(rrtx_handle* h, int32_t instance, double* x, double* y, double* cost, int32_t* parent, int32_t cap,
                  int32_t* n_out) {
  if (!h || instance < 0 || instance >= h->n_inst || !n_out) return RRTX_E_INVALID;
  if (!h->planned) return RRTX_E_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  Result r;
  HIPCHK(h, hipMemcpy(&r, h->c.results + instance, sizeof(r), hipMemcpyDeviceToHost));
  *n_out = r.n_nodes;
  if ((x || y || cost || parent) && cap < r.n_nodes) return RRTX_E_CAPACITY;
  if (h->p.algo == RRTX_ALGO_BITSTAR) {
    // tree.vertices in insertion order: coordinates of the grid ids (rrt_08:115-135), g-scores, `nodes` parents
    const int n = r.n_nodes;
    std::vector<double> vid(n), vg(n), vpar(n);
    std::vector<int32_t> vh(n);
    const double* d = h->ba.dslab + (int64_t)instance * rppb::DSLAB + 3LL * rppb::SC + 3LL * rppb::LC;
    HIPCHK(h, hipMemcpy(vid.data(), d, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(vg.data(), d + rppb::VC, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(vpar.data(), d + 3LL * rppb::VC, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(vh.data(), h->ba.islab + (int64_t)instance * rppb::ISLAB, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    const rpp::BitCfg& bc = h->bcfg[instance];
    for (int i = 0; i < n; i++) {
      const double c1 = std::floor(vid[i] / bc.num_cells), c0 = std::floor((vid[i] - c1 * bc.num_cells) / 1);
      if (x) x[i] = bc.rand_min + 0.01 * c0;
      if (y) y[i] = bc.rand_min + 0.01 * c1;
      if (cost) cost[i] = vg[i];
      if (parent) {
        parent[i] = -1;
        if (vh[i])
          for (int j = 0; j < n; j++)
            if (vid[j] == vpar[i]) parent[i] = j;
      }
    }
    return RRTX_OK;
  }
  const int64_t off = (int64_t)instance * h->stride;
  if (x) HIPCHK(h, hipMemcpy(x, h->c.x + off, sizeof(double) * r.n_nodes, hipMemcpyDeviceToHost));
  if (y) HIPCHK(h, hipMemcpy(y, h->c.y + off, sizeof(double) * r.n_nodes, hipMemcpyDeviceToHost));
  if (cost) HIPCHK(h, hipMemcpy(cost, h->c.cost + off, sizeof(double) * r.n_nodes, hipMemcpyDeviceToHost));
  if (parent) HIPCHK(h, hipMemcpy(parent, h->c.parent + off, sizeof(int32_t) * r.n_nodes, hipMemcpyDeviceToHost));
  return RRTX_OK;
}

int rrtx_get_path(rrtx_handle* h, int32_t instance, double* xy, int32_t cap_points, int32_t* n_out) {
  if (!h || instance < 0 || instance >= h->n_inst || !n_out) return RRTX_E_INVALID;
  if (!h->planned) return RRTX_E_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  Inst I;
  HIPCHK(h, hipMemcpy(&I, h->c.inst + instance, sizeof(I), hipMemcpyDeviceToHost));
  if (!(I.status & RRTX_ST_PATH)) {
    *n_out = 0;
    return RRTX_OK;
  }
  if (h->p.algo == RRTX_ALGO_BITSTAR) {
    int32_t oi[8];
    HIPCHK(h, hipMemcpy(oi, h->ba.out_i + 8 * instance, sizeof(oi), hipMemcpyDeviceToHost));
    *n_out = oi[3];
    if (!xy || oi[3] == 0) return RRTX_OK;
    if (cap_points < oi[3]) return RRTX_E_CAPACITY;
    const double* d = h->ba.dslab + (int64_t)instance * rppb::DSLAB + 3LL * rppb::SC + 3LL * rppb::LC + 5LL * rppb::VC +
                      2LL * rppb::EC;
    HIPCHK(h, hipMemcpy(xy, d, sizeof(double) * 2 * oi[3], hipMemcpyDeviceToHost));
    return RRTX_OK;
  }
  if (is_pose_tree(h->p.algo)) {
    // generate_final_course (rrt_05:1512-1521): [goal] + reversed edge polylines up the parent chain + [start]
    const int n = I.n;
    const int64_t off = (int64_t)instance * h->stride;
    std::vector<int32_t> par(n), plen(n);
    std::vector<int64_t> poff(n);
    HIPCHK(h, hipMemcpy(par.data(), h->c.parent + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(plen.data(), h->da.plen + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(poff.data(), h->da.poff + off, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
    int64_t total = 2;
    for (int nd = I.goal_node; par[nd] >= 0; nd = par[nd]) total += plen[nd];
    *n_out = (int32_t)total;
    if (!xy) return RRTX_OK;
    if (cap_points < total) return RRTX_E_CAPACITY;
    int64_t k = 0;
    xy[0] = I.goal[0];
    xy[1] = I.goal[1];
    k = 1;
    std::vector<double> bx, by;
    for (int nd = I.goal_node; par[nd] >= 0; nd = par[nd]) {
      bx.resize(plen[nd]);
      by.resize(plen[nd]);
      HIPCHK(h, hipMemcpy(bx.data(), h->pool_loc[instance].px + h->pool_loc[instance].slab * h->pool_loc[instance].cap + poff[nd],
                          sizeof(double) * plen[nd], hipMemcpyDeviceToHost));
      HIPCHK(h, hipMemcpy(by.data(), h->pool_loc[instance].py + h->pool_loc[instance].slab * h->pool_loc[instance].cap + poff[nd],
                          sizeof(double) * plen[nd], hipMemcpyDeviceToHost));
      for (int q = plen[nd] - 1; q >= 0; q--) {
        xy[2 * k] = bx[q];
        xy[2 * k + 1] = by[q];
        k++;
      }
    }
    xy[2 * k] = I.start[0];
    xy[2 * k + 1] = I.start[1];
    return RRTX_OK;
  }
  *n_out = I.path_n;
  if (!xy) return RRTX_OK;
  if (cap_points < I.path_n) return RRTX_E_CAPACITY;
  if (!(I.status & RRTX_ST_PATH_TRUNC)) {
    HIPCHK(h, hipMemcpy(xy, h->c.path_xy + (int64_t)instance * h->c.path_cap * 2, sizeof(double) * 2 * I.path_n,
                        hipMemcpyDeviceToHost));
    return RRTX_OK;
  }
  // deeper than the on-device path buffer: walk the parent array on the host (rrt_04:1117-1125)
  const int n = I.n;
  std::vector<double> x(n), y(n);
  std::vector<int32_t> par(n);
  const int64_t off = (int64_t)instance * h->stride;
  HIPCHK(h, hipMemcpy(x.data(), h->c.x + off, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(y.data(), h->c.y + off, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(par.data(), h->c.parent + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  int k = 0;
  xy[0] = I.goal[0];
  xy[1] = I.goal[1];
  k = 1;
  for (int nd = I.goal_node;; nd = par[nd]) {
    xy[2 * k] = x[nd];
    xy[2 * k + 1] = y[nd];
    k++;
    if (par[nd] < 0) break;
  }
  return RRTX_OK;
}

int rrtx_get_results(rrtx_handle* h, double* path_cost, int32_t* n_nodes, int32_t* status) {
  if (!h) return RRTX_E_INVALID;
  if (!h->planned && h->run.stage == 0) return RRTX_E_STATE;   // between steps of a plan: the records of the finished instances are final
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<Result> r(h->n_inst);
  HIPCHK(h, hipMemcpy(r.data(), h->c.results, sizeof(Result) * h->n_inst, hipMemcpyDeviceToHost));
  for (int i = 0; i < h->n_inst; i++) {
    if (path_cost) path_cost[i] = (r[i].status & RRTX_ST_PATH) ? r[i].path_cost : INFINITY;
    if (n_nodes) n_nodes[i] = r[i].n_nodes;
    if (status) status[i] = r[i].status;
  }
  return RRTX_OK;
}

int rrtx_results_device_ptr(rrtx_handle* h, void** dptr, int64_t* bytes) {
  if (!h || !dptr || !bytes) return RRTX_E_INVALID;
  *dptr = (void*)h->c.results;
  *bytes = (int64_t)sizeof(Result) * h->n_inst;
  return RRTX_OK;
}

int rrtx_copy_results_device(rrtx_handle* h, void* dst_device, int64_t bytes) {
  if (!h || !dst_device) return RRTX_E_INVALID;
  if (!h->planned) return RRTX_E_STATE;
  if (bytes < (int64_t)sizeof(Result) * h->n_inst) return RRTX_E_CAPACITY;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(dst_device, h->c.results, sizeof(Result) * h->n_inst, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RRTX_OK;
}

int rrtx_get_yaw(rrtx_handle* h, int32_t instance, double* yaw, int32_t cap) {
  if (!h || !yaw || instance < 0 || instance >= h->n_inst) return RRTX_E_INVALID;
  if (!h->planned || !is_pose_tree(h->p.algo)) return RRTX_E_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  Result r;
  HIPCHK(h, hipMemcpy(&r, h->c.results + instance, sizeof(r), hipMemcpyDeviceToHost));
  if (cap < r.n_nodes) return RRTX_E_CAPACITY;
  HIPCHK(h, hipMemcpy(yaw, h->da.yaw + (int64_t)instance * h->stride, sizeof(double) * r.n_nodes, hipMemcpyDeviceToHost));
  return RRTX_OK;
}

int rrtx_get_path_yaw(rrtx_handle* h, int32_t instance, double* yaw, int32_t cap_points, int32_t* n_out) {
  if (!h || instance < 0 || instance >= h->n_inst || !n_out) return RRTX_E_INVALID;
  if (!h->planned || h->p.algo != RRTX_ALGO_RS) return RRTX_E_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  Inst I;
  HIPCHK(h, hipMemcpy(&I, h->c.inst + instance, sizeof(I), hipMemcpyDeviceToHost));
  *n_out = 0;
  if (!(I.status & RRTX_ST_PATH)) return RRTX_OK;
  // generate_final_course (rrt_06:1643-1651): [goal yaw] + reversed edge-polyline yaws up the parent chain + [start yaw]
  const int n = I.n;
  const int64_t off = (int64_t)instance * h->stride;
  std::vector<int32_t> par(n), plen(n);
  std::vector<int64_t> poff(n);
  HIPCHK(h, hipMemcpy(par.data(), h->c.parent + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(plen.data(), h->da.plen + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(poff.data(), h->da.poff + off, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
  int64_t total = 2;
  for (int nd = I.goal_node; par[nd] >= 0; nd = par[nd]) total += plen[nd];
  *n_out = (int32_t)total;
  if (!yaw) return RRTX_OK;
  if (cap_points < total) return RRTX_E_CAPACITY;
  int64_t k = 0;
  yaw[k++] = I.goal[2];
  std::vector<double> bw;
  for (int nd = I.goal_node; par[nd] >= 0; nd = par[nd]) {
    bw.resize(plen[nd]);
    HIPCHK(h, hipMemcpy(bw.data(), h->pool_loc[instance].pyaw + h->pool_loc[instance].slab * h->pool_loc[instance].cap + poff[nd],
                        sizeof(double) * plen[nd], hipMemcpyDeviceToHost));
    for (int q = plen[nd] - 1; q >= 0; q--) yaw[k++] = bw[q];
  }
  yaw[k] = I.start[2];
  return RRTX_OK;
}

int rrtx_get_polylines(rrtx_handle* h, int32_t instance, int32_t* plen, int32_t cap_nodes, double* px, double* py,
                       int64_t cap_points, int64_t* n_points_out) {
  if (!h || !n_points_out || instance < 0 || instance >= h->n_inst) return RRTX_E_INVALID;
  if (!h->planned || !is_pose_tree(h->p.algo)) return RRTX_E_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  Result r;
  HIPCHK(h, hipMemcpy(&r, h->c.results + instance, sizeof(r), hipMemcpyDeviceToHost));
  const int n = r.n_nodes;
  const int64_t off = (int64_t)instance * h->stride;
  std::vector<int32_t> pl(n);
  std::vector<int64_t> po(n);
  HIPCHK(h, hipMemcpy(pl.data(), h->da.plen + off, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(po.data(), h->da.poff + off, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
  int64_t total = 0;
  for (int i = 0; i < n; i++) total += pl[i];
  *n_points_out = total;
  if (!plen && !px && !py) return RRTX_OK;
  if (cap_nodes < n || cap_points < total) return RRTX_E_CAPACITY;
  int64_t used = 0;
  HIPCHK(h, hipMemcpy(&used, h->da.pool_used + instance, sizeof(int64_t), hipMemcpyDeviceToHost));
  std::vector<double> bx(used), by(used);
  if (used) {
    const rrtx_handle::PoolLoc& pl = h->pool_loc[instance];
    HIPCHK(h, hipMemcpy(bx.data(), pl.px + pl.slab * pl.cap, sizeof(double) * used, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(by.data(), pl.py + pl.slab * pl.cap, sizeof(double) * used, hipMemcpyDeviceToHost));
  }
  int64_t w = 0;
  for (int i = 0; i < n; i++) {
    if (plen) plen[i] = pl[i];
    for (int q = 0; q < pl[i]; q++) {
      if (px) px[w] = bx[po[i] + q];
      if (py) py[w] = by[po[i] + q];
      w++;
    }
  }
  return RRTX_OK;
}

int rrtx_get_sobol_index(rrtx_handle* h, int32_t instance, int64_t* index) {
  if (!h || !index || instance < 0 || instance >= h->n_inst) return RRTX_E_INVALID;
  if (!h->planned) return RRTX_E_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  rpp::Sobol s;
  HIPCHK(h, hipMemcpy(&s, &h->c.inst[instance].sobol, sizeof(s), hipMemcpyDeviceToHost));
  *index = s.index;
  return RRTX_OK;
}

int rrtx_get_stats(rrtx_handle* h, rrtx_stats* st) {
  if (!h || !st) return RRTX_E_INVALID;
  *st = h->stats;
  return RRTX_OK;
}

int rrtx_get_phase_cycles(rrtx_handle* h, int64_t* out16) {
  if (!h || !out16) return RRTX_E_INVALID;
  memcpy(out16, h->phase, sizeof(h->phase));
  return RRTX_OK;
}

int rrtx_get_trace(rrtx_handle* h, double* rnd_x, double* rnd_y, int32_t* nearest, int32_t* n_near, int32_t cap,
                   int32_t* n_out) {
  if (!h || !n_out) return RRTX_E_INVALID;
  if (!h->planned || h->trace_inst < 0) return RRTX_E_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  if (h->p.algo == RRTX_ALGO_BITSTAR) {   // rnd_x / rnd_y carry the ids of the popped edges (bestEdge[0], bestEdge[1])
    int32_t oi[8];
    HIPCHK(h, hipMemcpy(oi, h->ba.out_i + 8 * h->trace_inst, sizeof(oi), hipMemcpyDeviceToHost));
    *n_out = oi[6];
    if (cap < oi[6] || oi[6] > h->ba.tr_cap) return RRTX_E_CAPACITY;
    if (rnd_x) HIPCHK(h, hipMemcpy(rnd_x, h->ba.tr_a, sizeof(double) * oi[6], hipMemcpyDeviceToHost));
    if (rnd_y) HIPCHK(h, hipMemcpy(rnd_y, h->ba.tr_b, sizeof(double) * oi[6], hipMemcpyDeviceToHost));
    return RRTX_OK;
  }
  Inst I;
  HIPCHK(h, hipMemcpy(&I, h->c.inst + h->trace_inst, sizeof(I), hipMemcpyDeviceToHost));
  const int n = I.it;
  *n_out = n;
  if (cap < n) return RRTX_E_CAPACITY;
  if (rnd_x) HIPCHK(h, hipMemcpy(rnd_x, h->c.tr_rx, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (rnd_y) HIPCHK(h, hipMemcpy(rnd_y, h->c.tr_ry, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (nearest) HIPCHK(h, hipMemcpy(nearest, h->c.tr_near, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  if (n_near) HIPCHK(h, hipMemcpy(n_near, h->c.tr_nn, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  return RRTX_OK;
}

int rrtx_get_trace_kind(rrtx_handle* h, int32_t* kind, int32_t cap, int32_t* n_out) {
  if (!h || !n_out) return RRTX_E_INVALID;
  if (!h->planned || h->trace_inst < 0 || !h->c.tr_kind) return RRTX_E_STATE;
  if (h->p.algo != RRTX_ALGO_RRT && h->p.algo != RRTX_ALGO_RRT_STAR) return RRTX_E_STATE;
  HIPCHK(h, hipSetDevice(h->device));
  Inst I;
  HIPCHK(h, hipMemcpy(&I, h->c.inst + h->trace_inst, sizeof(I), hipMemcpyDeviceToHost));
  *n_out = I.it;
  if (cap < I.it) return RRTX_E_CAPACITY;
  if (kind) HIPCHK(h, hipMemcpy(kind, h->c.tr_kind, sizeof(int32_t) * I.it, hipMemcpyDeviceToHost));
  return RRTX_OK;
}

// ---- path smoothing (rrt_04:1447-1479)
static int smooth_status_rc(const std::vector<int32_t>& st, std::string* err) {
  for (int32_t v : st) {
    if (v == rpps::SM_CAPACITY) {
      if (err) *err = "path smoothing: polyline or obstacle list exceeds the on-device capacity";
      return RRTX_E_OVERFLOW;
    }
    if (v == rpps::SM_ZERODIV) {
      if (err) *err = "path smoothing: the reference raises ZeroDivisionError on this input (zero-length pair)";
      return RRTX_E_STATE;
    }
  }
  return RRTX_OK;
}

int rrtx_smooth_paths(int32_t device, int32_t n_jobs, const double* paths_xy, const int32_t* path_n, int32_t in_stride,
                      int32_t max_iter, const double* obst_xyr, int32_t m, uint32_t* mt_words, int32_t* mt_pos,
                      double* out_xy, int32_t out_stride, int32_t* out_n, int32_t* status) {
  if (n_jobs < 1 || !paths_xy || !path_n || in_stride < 1 || max_iter < 0 || m < 0 || (m && !obst_xyr) || !mt_words ||
      !mt_pos || !out_xy || out_stride < 1 || !out_n || !status || m > rpps::MOB)
    return RRTX_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return RRTX_E_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return RRTX_E_HIP;
  std::vector<rpp::MT> rng(n_jobs);
  for (int j = 0; j < n_jobs; j++) {
    memcpy(rng[j].mt, mt_words + (size_t)j * 624, 624 * 4);
    rng[j].pos = mt_pos[j];
  }
  std::vector<double> ox(m + 1), oy(m + 1), osz(m + 1);
  for (int k = 0; k < m; k++) {
    ox[k] = obst_xyr[3 * k];
    oy[k] = obst_xyr[3 * k + 1];
    osz[k] = obst_xyr[3 * k + 2];
  }
  double *d_in = nullptr, *d_out = nullptr, *d_ox = nullptr, *d_oy = nullptr, *d_osz = nullptr;
  int32_t *d_n = nullptr, *d_on = nullptr, *d_st = nullptr;
  rpp::MT* d_rng = nullptr;
  int rc = RRTX_OK;
  auto A = [&](void** q, size_t bytes) { if (rc == RRTX_OK && hipMalloc(q, bytes ? bytes : 8) != hipSuccess) rc = RRTX_E_HIP; };
  A((void**)&d_in, sizeof(double) * 2 * (size_t)in_stride * n_jobs);
  A((void**)&d_out, sizeof(double) * 2 * (size_t)out_stride * n_jobs);
  A((void**)&d_ox, sizeof(double) * (m + 1));
  A((void**)&d_oy, sizeof(double) * (m + 1));
  A((void**)&d_osz, sizeof(double) * (m + 1));
  A((void**)&d_n, sizeof(int32_t) * n_jobs);
  A((void**)&d_on, sizeof(int32_t) * n_jobs);
  A((void**)&d_st, sizeof(int32_t) * n_jobs);
  A((void**)&d_rng, sizeof(rpp::MT) * n_jobs);
  std::vector<int32_t> st(n_jobs, 0);
  if (rc == RRTX_OK) {
    bool ok = hipMemcpy(d_in, paths_xy, sizeof(double) * 2 * (size_t)in_stride * n_jobs, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_ox, ox.data(), sizeof(double) * (m + 1), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_oy, oy.data(), sizeof(double) * (m + 1), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_osz, osz.data(), sizeof(double) * (m + 1), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_n, path_n, sizeof(int32_t) * n_jobs, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_rng, rng.data(), sizeof(rpp::MT) * n_jobs, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
      rpps::SmoothArgs a{d_in, in_stride, d_n, 1, d_rng, (int64_t)sizeof(rpp::MT), d_ox, d_oy, d_osz, m, max_iter,
                         d_out, out_stride, d_on, d_st};
      hipLaunchKernelGGL(rpps::smooth_kernel, dim3(n_jobs), dim3(64), 0, 0, a, n_jobs);
      ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
           hipMemcpy(out_xy, d_out, sizeof(double) * 2 * (size_t)out_stride * n_jobs, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(out_n, d_on, sizeof(int32_t) * n_jobs, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(st.data(), d_st, sizeof(int32_t) * n_jobs, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(rng.data(), d_rng, sizeof(rpp::MT) * n_jobs, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) rc = RRTX_E_HIP;
  }
  for (void* q : {(void*)d_in, (void*)d_out, (void*)d_ox, (void*)d_oy, (void*)d_osz, (void*)d_n, (void*)d_on, (void*)d_st,
                  (void*)d_rng})
    if (q) hipFree(q);
  if (rc != RRTX_OK) return rc;
  for (int j = 0; j < n_jobs; j++) {
    memcpy(mt_words + (size_t)j * 624, rng[j].mt, 624 * 4);
    mt_pos[j] = rng[j].pos;
    status[j] = st[j];
  }
  return smooth_status_rc(st, nullptr);
}

int rrtx_smooth_planned(rrtx_handle* h, int32_t max_iter) {
  if (!h || max_iter < 0) return RRTX_E_INVALID;
  if (!h->planned || (h->p.algo != RRTX_ALGO_RRT && h->p.algo != RRTX_ALGO_RRT_STAR)) return RRTX_E_STATE;
  if (h->m > rpps::MOB) return RRTX_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  const int B = h->n_inst;
  int rc;
  if (!h->sm_osz) {
    h->sm_stride = rpps::PC;
    if ((rc = dalloc(h, &h->sm_osz, rppk::MAX_OBS))) return rc;
    if ((rc = dalloc(h, &h->sm_xy, (size_t)2 * h->sm_stride * B))) return rc;
    if ((rc = dalloc(h, &h->sm_n, B))) return rc;
    if ((rc = dalloc(h, &h->sm_status, B))) return rc;
  }
  std::vector<double> osz(rppk::MAX_OBS, 0.0);
  for (int k = 0; k < h->m; k++) osz[k] = h->obst[3 * k + 2];
  HIPCHK(h, hipMemcpyAsync(h->sm_osz, osz.data(), sizeof(double) * rppk::MAX_OBS, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));   // `osz` is a local; the kernel below is ordered after it anyway
  Ctx& c = h->c;
  rpps::SmoothArgs a{c.path_xy, c.path_cap, &c.inst[0].path_n, (int64_t)(sizeof(Inst) / sizeof(int32_t)),
                     &c.inst[0].rng, (int64_t)sizeof(Inst), c.ox, c.oy, h->sm_osz, h->m, max_iter,
                     h->sm_xy, h->sm_stride, h->sm_n, h->sm_status};
  hipLaunchKernelGGL(rpps::smooth_kernel, dim3(B), dim3(64), 0, h->stream, a, B);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<int32_t> st(B);
  HIPCHK(h, hipMemcpy(st.data(), h->sm_status, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
  h->smoothed = true;
  return smooth_status_rc(st, &h->err);
}

int rrtx_get_smoothed_path(rrtx_handle* h, int32_t instance, double* xy, int32_t cap_points, int32_t* n_out) {
  if (!h || !n_out || instance < 0 || instance >= h->n_inst) return RRTX_E_INVALID;
  if (!h->smoothed) return RRTX_E_STATE;
  int32_t n = 0;
  HIPCHK(h, hipMemcpy(&n, h->sm_n + instance, sizeof(n), hipMemcpyDeviceToHost));
  *n_out = n;
  if (!xy || n == 0) return RRTX_OK;
  if (cap_points < n) return RRTX_E_CAPACITY;
  HIPCHK(h, hipMemcpy(xy, h->sm_xy + (size_t)2 * h->sm_stride * instance, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
  return RRTX_OK;
}

int rrtx_selftest_math(int32_t device, int32_t op, const double* a, const double* b, double* out, int64_t n) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RRTX_E_NO_DEVICE;
  if (!a || !b || !out || n < 0) return RRTX_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return RRTX_E_HIP;
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  int rc = RRTX_OK;
  if (hipMalloc(&da, n * 8) != hipSuccess || hipMalloc(&db, n * 8) != hipSuccess ||
      hipMalloc(&dout, n * 8) != hipSuccess)
    rc = RRTX_E_HIP;
  if (!rc && (hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice) != hipSuccess ||
              hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice) != hipSuccess))
    rc = RRTX_E_HIP;
  if (!rc) {
    hipLaunchKernelGGL(rppk::selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, da, db, dout, n);
    if (hipDeviceSynchronize() != hipSuccess) rc = RRTX_E_HIP;
  }
  if (!rc && hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = RRTX_E_HIP;
  hipFree(da);
  hipFree(db);
  hipFree(dout);
  return rc;
}

int rrtx_selfcheck(int32_t device, int32_t n_per_fn, int64_t* mismatches8) {
  if (!mismatches8 || n_per_fn < 1 || n_per_fn > (1 << 22)) return RRTX_E_INVALID;
  const int64_t n = n_per_fn;
  std::vector<double> a(n), b(n), out(n);
  // splitmix64 -> uniform in [0, 1): the same arguments on every host
  uint64_t sm = 0x9e3779b97f4a7c15ULL;
  auto u01 = [&]() {
    uint64_t z = (sm += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
  };
  // selftest op, argument ranges: coordinates differences up to a few hundred, angles within a few turns (Dubins /
  // Reeds-Shepp sums, 2*pi*a/b of the unit-ball sample), |x| <= 1 for the inverse functions
  struct Fn { int op; double lo_a, hi_a, lo_b, hi_b; } fns[8] = {
      {1, -300.0, 300.0, 0.0, 1.0}, {2, -20.0, 20.0, 0.0, 1.0}, {3, -20.0, 20.0, 0.0, 1.0}, {4, -300.0, 300.0, -300.0, 300.0},
      {8, -1.0, 1.0, 0.0, 1.0},     {9, -1.0, 1.0, 0.0, 1.0},   {6, 0.0, 1.0e5, 0.0, 1.0},  {7, -300.0, 300.0, -300.0, 300.0}};
  for (int f = 0; f < 8; f++) {
    for (int64_t i = 0; i < n; i++) {
      a[i] = fns[f].lo_a + (fns[f].hi_a - fns[f].lo_a) * u01();
      b[i] = fns[f].lo_b + (fns[f].hi_b - fns[f].lo_b) * u01();
      if (i % 7 == 3 && (fns[f].op == 1 || fns[f].op == 4)) a[i] *= 1.0 / 1024.0;   // small arguments too
    }
    int rc = rrtx_selftest_math(device, fns[f].op, a.data(), b.data(), out.data(), n);
    if (rc) return rc;
    int64_t bad = 0;
    for (int64_t i = 0; i < n; i++) {
      double r;
      switch (fns[f].op) {
        case 1: r = py_sq_host(a[i]); break;
        case 2: r = libm_sin(a[i]); break;
        case 3: r = libm_cos(a[i]); break;
        case 4: r = libm_atan2(a[i], b[i]); break;
        case 8: r = libm_acos(a[i]); break;
        case 9: r = libm_asin(a[i]); break;
        case 6: r = libm_sqrt(a[i]); break;
        default: r = a[i] / b[i]; break;
      }
      if (memcmp(&r, &out[i], 8) != 0) bad++;
    }
    mismatches8[f] = bad;
  }
  return RRTX_OK;
}

// ---- native RCCL: the one collective of the path (ncclAllGather of the 16-byte result records over xGMI) ----------

int rrtx_rccl_unique_id(void* id128) {
  if (!id128) return RRTX_E_INVALID;
  RcclApi* a = rccl_api();
  if (!a->lib) return RRTX_E_STATE;
  ncclUniqueId id;
  if (a->GetUniqueId(&id) != ncclSuccess) return RRTX_E_HIP;
  memcpy(id128, &id, sizeof(id));
  return RRTX_OK;
}

int rrtx_rccl_init(rrtx_handle* h, const void* id128, int32_t rank, int32_t world) {
  if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return RRTX_E_INVALID;
  RcclApi* a = rccl_api();
  if (!a->lib) {
    h->err = a->err;
    return RRTX_E_STATE;
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (h->rccl_comm) {
    a->CommDestroy((ncclComm_t)h->rccl_comm);
    h->rccl_comm = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = a->CommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) {
    h->err = std::string("ncclCommInitRank: ") + (a->GetErrorString ? a->GetErrorString(r) : "error");
    return RRTX_E_HIP;
  }
  h->rccl_comm = (void*)comm;
  h->rccl_world = world;
  h->rccl_rank = rank;
  if (!h->rccl_recv) {
    int rc = dalloc(h, &h->rccl_recv, (size_t)h->n_inst * world);
    if (rc) return rc;
  }
  return RRTX_OK;
}

int rrtx_rccl_gather_results(rrtx_handle* h, double* path_cost, int32_t* n_nodes, int32_t* status) {
  if (!h) return RRTX_E_INVALID;
  if (!h->planned || !h->rccl_comm) return RRTX_E_STATE;
  RcclApi* a = rccl_api();
  HIPCHK(h, hipSetDevice(h->device));
  const size_t bytes = sizeof(Result) * (size_t)h->n_inst;
  // device -> device: the table the planner kernels wrote is what the collective sends
  const ncclResult_t r = a->AllGather(h->c.results, h->rccl_recv, bytes, ncclInt8, (ncclComm_t)h->rccl_comm, h->stream);
  if (r != ncclSuccess) {
    h->err = std::string("ncclAllGather: ") + (a->GetErrorString ? a->GetErrorString(r) : "error");
    return RRTX_E_HIP;
  }
  const size_t tot = (size_t)h->n_inst * h->rccl_world;
  std::vector<Result> all(tot);
  HIPCHK(h, hipMemcpyAsync(all.data(), h->rccl_recv, sizeof(Result) * tot, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < tot; i++) {
    if (path_cost) path_cost[i] = (all[i].status & RRTX_ST_PATH) ? all[i].path_cost : INFINITY;
    if (n_nodes) n_nodes[i] = all[i].n_nodes;
    if (status) status[i] = all[i].status;
  }
  return RRTX_OK;
}

int rrtx_plan_many(rrtx_handle** handles, int32_t n, int32_t* rcs) {
  if (!handles || n < 1) return RRTX_E_INVALID;
  for (int i = 0; i < n; i++)
    if (!handles[i]) return RRTX_E_INVALID;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++)
      if (handles[i] == handles[j]) return RRTX_E_INVALID;   // a handle is not thread safe
  std::vector<int> rc(n, RRTX_OK);
  if (n == 1) {
    rc[0] = rrtx_plan(handles[0]);
  } else {
    std::vector<std::thread> th;
    th.reserve(n);
    for (int i = 0; i < n; i++) th.emplace_back([&rc, handles, i]() { rc[i] = rrtx_plan(handles[i]); });
    for (auto& t : th) t.join();
  }
  int worst = RRTX_OK;
  for (int i = 0; i < n; i++) {
    if (rcs) rcs[i] = rc[i];
    if (rc[i] < 0 && (worst >= 0 || rc[i] < worst)) worst = rc[i];
    else if (rc[i] == RRTX_PARTIAL && worst == RRTX_OK) worst = RRTX_PARTIAL;
  }
  return worst;
}

}  // extern "C"
