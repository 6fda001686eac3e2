/*
 * rrtx.h -- C ABI of the MI355X-native batched RRT / RRT* planner.
 *
 * The reference (gouldberg/robotics-path-planning) has no FFI layer: its
 * boundary is the Python class surface `RRT(...).planning(animation)` and the
 * attributes callers read afterwards.  This ABI is what a ctypes
 * binding on the reference side calls in place of the body of
 *   10_path_planning_01_rrt_01_simple.py   RRT.planning            :71-101
 *   10_path_planning_01_rrt_04_rrt_star.py RRT.planning            :1036-1084
 *   10_path_planning_01_rrt_07_informed_rrt_star.py RRT.informed_rrt_star_search :1044-1108
 *   10_path_planning_01_rrt_05_rrt_star_dubins_path.py RRT.planning :1416-1456
 *   10_path_planning_01_rrt_08_batch_informed_rrt_star.py BITStar.plan :236-331
 * Each entry point names the reference interface it replaces.  Plain pointers
 * and sizes only; the caller owns every host buffer, the library owns device
 * memory behind the opaque handle.  Every function returns 0 or a negative
 * RRTX_E_* code and never throws.  A handle is bound to one device and is not
 * thread safe (one host thread per handle; multi-GPU = one handle per device).
 * There is NO CPU fallback: without a usable gfx950 device rrtx_create fails
 * with RRTX_E_NO_DEVICE.
 */
#ifndef RRTX_H
#define RRTX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RRTX_ABI_VERSION 5   /* 5: rrtx_stats.passes_shared; 2: rrtx_params.step_size, RRTX_ALGO_RS, rrtx_get_path_yaw; 3: RRTX_PARTIAL,
                                rrtx_copy_results_device, per-instance yaw and informed rotation; 4: rrtx_plan_many,
                                rrtx_selfcheck, rrtx_stats.main_shape / main_f32, rrtx_plan_begin / _step, rrtx_set_launch_bound,
                                RRTX_ST_REF_HANGS, rrtx_rccl_* */

enum {
  RRTX_PARTIAL = 1,        /* rrtx_plan only: the call completed, but at least one instance stopped with RRTX_ST_OVERFLOW,
                              RRTX_ST_UNSUPPORTED or RRTX_ST_REF_RAISES in its status word (rrtx_get_results); every
                              other instance is complete and valid.  Not an error: errors are negative. */
  RRTX_OK = 0,
  RRTX_E_INVALID = -1,     /* bad argument / unsupported parameter combination */
  RRTX_E_NO_DEVICE = -2,   /* no HIP device, or not gfx950 */
  RRTX_E_HIP = -3,         /* HIP runtime error, see rrtx_last_error */
  RRTX_E_CAPACITY = -4,    /* caller buffer too small */
  RRTX_E_STATE = -5,       /* call order (e.g. get_tree before plan) */
  RRTX_E_OVERFLOW = -6     /* a whole-call capacity was exceeded (path smoothing); per-instance overflows of rrtx_plan are
                              reported as RRTX_PARTIAL + RRTX_ST_OVERFLOW */
};

enum { RRTX_ALGO_RRT = 0,       /* rrt_01 RRT.planning :71-101 */
       RRTX_ALGO_RRT_STAR = 1,  /* rrt_04 RRT.planning :1036-1084 */
       RRTX_ALGO_INFORMED = 2,  /* rrt_07 RRT.informed_rrt_star_search :1044-1108 */
       RRTX_ALGO_DUBINS = 3,    /* rrt_05 RRT.planning :1416-1456 (RRT*-Dubins; start[2]/goal[2] = yaw) */
       RRTX_ALGO_BITSTAR = 4,   /* rrt_08 BITStar.plan :236-331 (max_iter = maxIter; rand_area = randArea) */
       RRTX_ALGO_RRT_DUBINS = 5 /* rrt_03 RRT.planning :1420-1456 (RRT with Dubins steer; poses, curvature and goal
                                   thresholds as RRTX_ALGO_DUBINS; sampler SOBOL = the 3-D point of :1545-1563) */,
       RRTX_ALGO_RS = 6         /* rrt_06 RRT.planning :1530-1570 (RRT*-Reeds-Shepp incl. try_goal_path :1572-1582; poses,
                                   curvature, step_size and goal thresholds; node capacity 2 * max_iter + 2; the sampler is
                                   always get_random_node :1658-1666, as in the reference's loop :1539) */ };
enum { RRTX_SAMPLER_MT = 0,     /* get_random_node        rrt_04:1132-1139 */
       RRTX_SAMPLER_SOBOL = 1   /* get_random_node_sobol  rrt_04:1142-1153 */ };

/* per-instance status bits (rrtx_get_results) */
enum { RRTX_ST_DONE = 1, RRTX_ST_PATH = 2, RRTX_ST_OVERFLOW = 4, RRTX_ST_PATH_TRUNC = 8,
       RRTX_ST_UNSUPPORTED = 16 /* a reference code path the device kernel does not restate was reached; the instance
                                   stops there instead of continuing differently.  No planner sets it in a result any
                                   more: the one such path (rrt_04 rewire visiting a MOVED node again, :1337 with :1372)
                                   is walked by the general kernel, to which rrtx_plan hands such instances over
                                   (rrtx_stats.replanned); kept as a guard */,
       RRTX_ST_REF_RAISES = 32  /* RRTX_ALGO_RS: the reference raises here (ZeroDivisionError :1183/:1207 or ValueError from
                                   math.acos/asin) inside reeds_shepp_path_planning; the instance stops, no path */,
       RRTX_ST_REF_HANGS = 64   /* RRTX_ALGO_BITSTAR, set together with RRTX_ST_OVERFLOW: the reference does not terminate on
                                   this instance.  plan() adds samples only `if iterations != 0` (rrt_08:215); when no edge
                                   ever connects (a start walled in by obstacles: every connect() fails and `continue`s past
                                   the iteration counter, :283) both queues run dry a second time with the tree, the samples
                                   and the RNG unchanged, and the same round repeats for ever.  The device proves that at the
                                   second arrival and stops the instance there */ };

/* Constructor arguments of the reference classes (rrt_04:951-1000, rrt_01:32-69). */
typedef struct rrtx_params {
  int32_t abi_version;           /* RRTX_ABI_VERSION */
  int32_t algo;                  /* RRTX_ALGO_* */
  int32_t sampler;               /* sobol_sampler (rrt_04:962) */
  int32_t goal_sample_rate;      /* rrt_04:958 */
  int32_t max_iter;              /* rrt_04:959 */
  int32_t has_play_area;         /* play_area is not None (rrt_04:981) */
  int32_t search_until_max_iter; /* rrt_04:964 */
  int32_t n_instances;           /* independent planning instances resident on the device */
  int32_t device;                /* HIP device ordinal */
  int32_t reserved_i[7];
  double start[3];               /* start [x,y,(yaw)] rrt_04:977 */
  double goal[3];                /* goal  [x,y,(yaw)] rrt_04:978 */
  double rand_min, rand_max;     /* rand_area rrt_04:979-980 */
  double expand_dis;             /* rrt_04:985 */
  double path_resolution;        /* rrt_04:986 */
  double play_area[4];           /* xmin xmax ymin ymax (rrt_04:944-949) */
  double robot_radius;           /* rrt_04:991 */
  double connect_circle_dist;    /* rrt_04:998 */
  /* RRTX_ALGO_INFORMED only: upper-left 2x2 of the rotation `c` (numpy SVD, rrt_07:1061-1068), row major, and
   * c_min = math.hypot(start - goal) (rrt_07:1054); the host computes both exactly as the reference does. */
  double informed_rot[4];
  double informed_c_min;
  /* RRTX_ALGO_DUBINS / RRTX_ALGO_RRT_DUBINS only (rrt_05:1371-1373, 1411-1413; rrt_03:1381-1383, 1416-1418) */
  double curvature, goal_yaw_th, goal_xy_th;
  /* RRTX_ALGO_RS only: step_size of the Reeds-Shepp interpolation (rrt_06:1484, :1525) */
  double step_size;
  double reserved_d[3];
} rrtx_params;

/* Aggregate counters over all instances of the last rrtx_plan(). */
typedef struct rrtx_stats {
  int64_t iterations;        /* loop iterations executed (rrt_04:1044) */
  int64_t edges_unique;      /* collision-checked edge expansions actually evaluated on the device */
  int64_t edges_ref;         /* check_collision calls the reference would have made (repeated near indices counted) */
  int64_t near_hits;         /* sum of len(near_inds) (rrt_04:1337) */
  int64_t near_unique;       /* distinct indices among them */
  int64_t rewires;           /* rrt_04:1368 successes */
  int64_t propagated;        /* nodes rewritten by propagate_cost_to_leaves (rrt_04:1379-1384) */
  int64_t scan_nodes;        /* nodes visited by nearest/near/goal scans */
  int64_t algorithmic_bytes; /* bytes this implementation's algorithm must move: RRTX_ALGO_RRT_STAR 4*n per pass over the
                                16-bit coordinate mirror (8*n per f32-mirror pass, 16*n per f64 pass: the fallbacks,
                                RRTX_Q16=0 / RRTX_F32=0 and the other algorithms) + 16 per hit gathered + 48*k + 24*M +
                                44; the near pass of iteration i also serves the nearest query of i+1 */
  int64_t exact_rescans;     /* nearest scans re-done with exact ** 2 (tie within filter margin) */
  int64_t total_nodes;
  int64_t launches;          /* kernel launches */
  double kernel_ms;          /* HIP-event time of all planner-kernel launches on the handle's stream */
  double plan_ms;            /* host wall time of rrtx_plan */
  int64_t algorithmic_bytes_two_scan; /* SURVEY.md 8d formula as written: 32*n + 48*k + 24*M + 28 per accepted iteration */
  int64_t near_unique_max;   /* largest number of distinct near candidates any iteration of any instance produced */
  int64_t f32_fallbacks;     /* nearest queries the f32-mirror pass could not decide (repeated with the f64 pass) */
  int64_t q16_fallbacks;     /* nearest queries the 16-bit first stage could not decide on grid distances (decided by a
                                second 16-bit pass that collects the candidates + their f64 coordinates) */
  /* the DOMINANT kernel alone (RRTX_ALGO_RRT_STAR with search_until_max_iter: rrt_star_kernel_v2, whose launches a
   * rocprofv3 --kernel-trace --stats summary lists separately from the final goal-search launch; the other algorithms:
   * their one planner kernel = the totals above) */
  int64_t launches_main;
  double kernel_ms_main;
  int64_t replanned;         /* instances planned a second time from their staged start state: a near set that outgrew its
                                workgroup shape's table, a polyline pool that ran out, or an rrt_04 rewire that moved a
                                node while near_inds held repeated indices (raw-list walk of the general kernel) */
  int32_t main_shape;        /* threads per workgroup (= per planning instance) of the dominant kernel as launched:
                                RRTX_ALGO_RRT_STAR iteration kernel 64 / 128 / 256 (picked from the instance count, the
                                obstacle count, the estimated near-set size and RRTX_TPB); the other planners' fixed shape */
  int32_t main_f32;          /* RRTX_ALGO_RRT_STAR iteration kernel: 1 = f32-mirror instantiation (<true>), 0 = f64 passes */
  int64_t passes_shared;     /* RRTX_ALGO_RRT_STAR iteration kernel, 64-thread shape: iterations whose near query was answered
                                by the streaming pass of an earlier iteration (the ball speculated about the sample, on which
                                steer() snaps the new node once the tree is dense; up to three iterations ride on one pass):
                                no pass of their own, 0 bytes of xq[] */
} rrtx_stats;

typedef struct rrtx_handle rrtx_handle;

/* replaces RRT.__init__ (rrt_04:951-1000); allocates device state for n_instances trees of max_iter+1 nodes. */
int rrtx_create(const rrtx_params* p, rrtx_handle** out);
/* obstacle_list ctor argument (rrt_04:989): m rows of (ox, oy, size), AoS in; converted to SoA + thresholds
 * (size+robot_radius)**2 (rrt_04:1227) on the host.  Shared by all instances of the handle. */
int rrtx_set_obstacles(rrtx_handle* h, const double* oxyr, int32_t m);
/* hand over / take back CPython's `random.getstate()[1]` (624 words + position) for one instance, so that
 * the device consumes the stream exactly as random.randint / random.uniform would (rrt_04:1133-1136). */
int rrtx_set_rng_state(rrtx_handle* h, int32_t instance, const uint32_t* mt624, int32_t pos);
int rrtx_get_rng_state(rrtx_handle* h, int32_t instance, uint32_t* mt624, int32_t* pos);
/* convenience: state after `random.seed(seed)` for instances first..first+count-1 */
int rrtx_seed_instances(rrtx_handle* h, int32_t first, int32_t count, const uint64_t* seeds);
/* per-instance start / goal for batches (default: the ctor's): x, y and, for the pose planners (RRTX_ALGO_DUBINS /
 * _RRT_DUBINS / _RS), yaw in element 2 (rrt_05:1406-1407, rrt_06:1518-1519).  Either pointer may be NULL. */
int rrtx_set_instance(rrtx_handle* h, int32_t instance, const double* start3, const double* goal3);
/* Per-instance rotation `C` (upper-left 2x2, row major) and c_min, computed by the host with numpy exactly as the
 * reference does for that instance's start / goal (default: the ctor's informed_rot / informed_c_min):
 * RRTX_ALGO_BITSTAR cMin = hypot(start-goal)/1.5 and C of rrt_08:189-202; RRTX_ALGO_INFORMED c_min = hypot(start-goal)
 * and C of rrt_07:1054-1068 (the ellipse centre follows rrtx_set_instance). */
int rrtx_set_instance_rotation(rrtx_handle* h, int32_t instance, const double* rot4, double c_min);
/* replaces the body of RRT.planning(animation=False) for every instance; blocking.  Returns RRTX_OK, RRTX_PARTIAL
 * (see above) or a negative error. */
int rrtx_plan(rrtx_handle* h);
/* The same plan in BOUNDED launches (load balance / a service that must not wait for its slowest instance).
 * rrtx_plan_begin uploads the instances' start state; every rrtx_plan_step queues ONE kernel launch over the instances that
 * are not finished, waits for it and reports in *n_pending how many instances still need another launch (0 = the plan is
 * complete: the step that reaches 0 also runs the overflow re-plans and returns what rrtx_plan would have returned).  A
 * launch is bounded by rrtx_set_launch_bound (default: RRT* iteration kernel 131072 iterations, the other tree planners
 * 32768, BIT* 20000 trips of plan()'s loop :243): an instance that has used its share stores its state on the device and is
 * carried into the next launch -- results do not depend on the bound.  Between steps rrtx_get_results is valid: an instance
 * with RRTX_ST_DONE in its status word is final, whatever the others still do.  rrtx_plan = begin + steps until 0.
 * BIT* launches run persistent waves over a device-side work queue of the pending instances (rrt_bitstar_wave.hip.h). */
int rrtx_plan_begin(rrtx_handle* h);
int rrtx_plan_step(rrtx_handle* h, int32_t* n_pending);
/* iterations (BIT*: loop trips) one launch may spend on one instance; not while a plan is in progress */
int rrtx_set_launch_bound(rrtx_handle* h, int32_t iterations);
/* Multi-GPU in one process ("one handle per device, driven from one process with N threads"): plans the n
 * handles concurrently, one host thread per handle (each bound to its handle's device; two handles may share a device),
 * and returns when all have finished.  rcs[i] (may be NULL) = what rrtx_plan(handles[i]) returned; the return value is
 * the most severe of them (a negative error, else RRTX_PARTIAL, else RRTX_OK).  Instances are independent, so sharding a
 * batch over handles changes no result: instance j of handle i is the instance it would be in one large handle with the
 * same seed / start / goal.  The handles must be distinct. */
int rrtx_plan_many(rrtx_handle** handles, int32_t n, int32_t* rcs);
/* Multi-GPU with one process per GPU, without any host framework: the ONE collective of the path -- an ncclAllGather of the
 * 16-byte result records {f64 path_cost, i32 n_nodes, i32 status}, device to device over RCCL / xGMI.  RCCL is
 * opened with dlopen at first use (librrtx.so does not link it).  rank 0 calls rrtx_rccl_unique_id and hands the 128 bytes to
 * the other ranks by the host's own means (a file, MPI, a torch store); every rank calls rrtx_rccl_init on its handle (same
 * n_instances on every rank), plans, then rrtx_rccl_gather_results: world * n_instances entries, rank-major -- the table
 * rrtx_get_results gives for one rank, for all of them.  No reference counterpart (the reference is one process). */
int rrtx_rccl_unique_id(void* id128);
int rrtx_rccl_init(rrtx_handle* h, const void* id128, int32_t rank, int32_t world);
int rrtx_rccl_gather_results(rrtx_handle* h, double* path_cost, int32_t* n_nodes, int32_t* status);
/* rrt.node_list as SoA: x, y, cost (f64), parent (i32, -1 = None); any pointer may be NULL. */
int rrtx_get_tree(rrtx_handle* h, int32_t instance, double* x, double* y, double* cost, int32_t* parent,
                  int32_t cap, int32_t* n_out);
/* return value of planning(): n_out points [x,y] goal -> start (rrt_04:1117-1125); n_out = 0 <=> None. */
int rrtx_get_path(rrtx_handle* h, int32_t instance, double* xy, int32_t cap_points, int32_t* n_out);
/* per-instance result table {path_cost = get_path_length(path) (rrt_04:1391-1399), n_nodes, status};
 * this 16-byte record is what multi-GPU runs gather over RCCL. */
int rrtx_get_results(rrtx_handle* h, double* path_cost, int32_t* n_nodes, int32_t* status);
/* device pointer + byte size of the packed result table (n_instances x {f64 cost, i32 n, i32 status}) */
int rrtx_results_device_ptr(rrtx_handle* h, void** dptr, int64_t* bytes);
/* the same table copied device -> device into a caller-owned device buffer (e.g. the tensor an RCCL all_gather sends):
 * no round trip through host memory; `bytes` = capacity of dst_device (>= 16 * n_instances) */
int rrtx_copy_results_device(rrtx_handle* h, void* dst_device, int64_t bytes);
/* RRTX_ALGO_DUBINS: node yaw (rrt_05 Node.yaw) and the stored edge polylines (Node.path_x / path_y, :1472-1474):
 * plen[i] points per node, concatenated in node order into px/py. */
int rrtx_get_yaw(rrtx_handle* h, int32_t instance, double* yaw, int32_t cap);
/* RRTX_ALGO_RS: third column of generate_final_course (rrt_06:1643-1651), same points as rrtx_get_path */
int rrtx_get_path_yaw(rrtx_handle* h, int32_t instance, double* yaw, int32_t cap_points, int32_t* n_out);
int rrtx_get_polylines(rrtx_handle* h, int32_t instance, int32_t* plen, int32_t cap_nodes, double* px, double* py,
                       int64_t cap_points, int64_t* n_points_out);
/* Sobol index (RRT.sobol_inter_, rrt_04:995,1148) after planning */
int rrtx_get_sobol_index(rrtx_handle* h, int32_t instance, int64_t* index);
int rrtx_get_stats(rrtx_handle* h, rrtx_stats* st);
/* optional per-iteration trace of one instance (debug/parity harness): call before rrtx_plan.
 * rows: rnd_x, rnd_y (f64), nearest, n_near_unique (i32; -1 when the iteration stopped before the near query) */
int rrtx_enable_trace(rrtx_handle* h, int32_t instance);
int rrtx_get_trace(rrtx_handle* h, double* rnd_x, double* rnd_y, int32_t* nearest, int32_t* n_near,
                   int32_t cap, int32_t* n_out);
/* RRTX_ALGO_RRT / RRTX_ALGO_RRT_STAR, same rows: what the iteration appended -- 0 nothing, 1 the extension edge itself
 * (rrt_01:85-96; rrt_04:1066-1067, choose_parent returned None), 2 a node under a chosen parent (rrt_04:1062-1065).  With
 * rnd / nearest this lets the host rebuild Node.path_x / path_y exactly as the reference holds them (draw data). */
int rrtx_get_trace_kind(rrtx_handle* h, int32_t* kind, int32_t cap, int32_t* n_out);
/* diagnostic builds only (-DRRTX_PHASE_TIMERS): shader-clock cycles per kernel phase summed over instances
 * (0 sample, 1 nearest scan, 2 steer, 3 extension collision, 4 near scan, 5 exact re-check + de-dup,
 *  6 choose_parent edges, 7 choose_parent costs, 8 rewire edges, 9 rewire resolve + propagate + append,
 *  11 bookkeeping, 12 goal search, 15 loop overhead); all zero in the shipped build. */
int rrtx_get_phase_cycles(rrtx_handle* h, int64_t* out16);
const char* rrtx_last_error(rrtx_handle* h);
void rrtx_destroy(rrtx_handle* h);

/* library-level */
int rrtx_abi_version(void);
int rrtx_device_count(void);
/* Evaluate the device scalar core on arrays (parity harness for the glibc/CPython arithmetic replicas):
 * op 0: math.hypot(a,b)  1: a**2  2: math.sin(a)  3: math.cos(a)  4: math.atan2(a,b)
 * op 5: steer end x of (0,0)->(a,b) with extend=inf, res=0.25   6: sqrt(a)  7: a/b */
/* Path smoothing: replaces path_smoothing(path, max_iter, obstacle_list) (rrt_04:1447-1479; get_path_length :1391,
 * get_target_point :1401, line_collision_check :1423 -- the infinite-line distance test is kept), which every driver
 * runs right after planning() on the path it returned, drawing from the same MT19937 stream (rrt_04:1558-1559).
 *
 * rrtx_smooth_paths: a batch of polylines from the host.  Job j: path_n[j] points at paths_xy + 2*j*in_stride
 * (goal -> start order as planning() returns them), RNG state mt_words[624*j .. ], mt_pos[j] (advanced in place, so
 * random.getstate() afterwards equals the reference's); result out_n[j] points at out_xy + 2*j*out_stride;
 * status[j] = 0 ok, 1 capacity (more than 512 points / 256 obstacles), 2 the reference raises ZeroDivisionError.
 * obst_xyr = m rows (ox, oy, size) -- sizes as given, no robot radius (:1441). */
int rrtx_smooth_paths(int32_t device, int32_t n_jobs, const double* paths_xy, const int32_t* path_n, int32_t in_stride,
                      int32_t max_iter, const double* obst_xyr, int32_t m, uint32_t* mt_words, int32_t* mt_pos,
                      double* out_xy, int32_t out_stride, int32_t* out_n, int32_t* status);
/* The same on the paths a handle just planned (RRTX_ALGO_RRT / RRTX_ALGO_RRT_STAR), entirely on the device: every
 * instance's path is smoothed continuing that instance's RNG stream (rrtx_get_rng_state afterwards = after smoothing). */
int rrtx_smooth_planned(rrtx_handle* h, int32_t max_iter);
int rrtx_get_smoothed_path(rrtx_handle* h, int32_t instance, double* xy, int32_t cap_points, int32_t* n_out);

/* parity harness: out[i] = op(a[i], b[i]) evaluated on the device.  op 0 math.hypot, 1 x**2, 2 sin, 3 cos, 4 atan2,
 * 5 steer end x (rrt_04:1086-1115), 6 sqrt, 7 a/b, 8 acos, 9 asin, 10 checksum of the Reeds-Shepp steer
 * (0,0,0) -> (a, b, a+b) (rrt_06:1426-1441, csrc/rpp_rs.h) */
int rrtx_selftest_math(int32_t device, int32_t op, const double* a, const double* b, double* out, int64_t n);

/* Run-time check of the arithmetic contract: "identical to the reference on this host" holds while
 * the host's libm -- the one the reference's CPython calls -- returns what the device's operation-by-operation replicas
 * of glibc 2.35's x86-64 FMA variants return.  Evaluates n_per_fn seeded arguments per function on the device and with
 * the HOST's libm, over the argument ranges the planners use, and counts the results that differ in any bit:
 * mismatches8[0..7] = pow(x, 2), sin, cos, atan2, acos, asin, sqrt, a / b.  Returns RRTX_OK when the check ran (whatever it
 * found).  All zero: doubles are bit-identical to the reference run on this host.  Otherwise the planners still run and
 * are self-consistent, but match the reference only up to the few-ULP differences between the two libm builds (integer
 * results can then differ at near-ties).  math.hypot is CPython's own algorithm, not libm's: the Python host checks it
 * (and float ** 2) against the interpreter itself with rrtx_selftest_math (robotics-path-planning_amd/_abi.py selfcheck). */
int rrtx_selfcheck(int32_t device, int32_t n_per_fn, int64_t* mismatches8);

#ifdef __cplusplus
}
#endif
#endif
