// rrt_star_v2.hip.h -- instantiates the latency-lean RRT* iteration kernel (rrt_star_v2_body.inc) for two
// workgroup shapes:
//   rppk2     256 threads / instance, 4 workgroups per CU  (<= 1024 resident instances per GPU)
//   rppk2s    128 threads / instance, 8 workgroups per CU  (<= 2048 resident instances; smaller LDS tables)
//   rppk2t     64 threads / instance, 16 workgroups per CU (<= 4096 resident instances; < 10 KB LDS each)
// More resident instances hide the serial phases' memory latency behind other instances' node-array streams.
#pragma once
#include "rrt_kernels.hip.h"

#define RRT2_NS rppk2
#define RRT2_TPB 256
#define RRT2_MAXOBS 256
#define RRT2_NU 256
#define RRT2_EBD 32
#define RRT2_HW 48
#define RRT2_FCAP 64
#define RRT2_WPS 4
#include "rrt_star_v2_body.inc"
#undef RRT2_NS
#undef RRT2_TPB
#undef RRT2_MAXOBS
#undef RRT2_NU
#undef RRT2_EBD
#undef RRT2_HW
#undef RRT2_FCAP
#undef RRT2_WPS

#define RRT2_NS rppk2s
#define RRT2_TPB 128
#define RRT2_MAXOBS 64
#define RRT2_NU 128
#define RRT2_EBD 16
#define RRT2_HW 24
#define RRT2_FCAP 32
#define RRT2_WPS 4
#include "rrt_star_v2_body.inc"
#undef RRT2_NS
#undef RRT2_TPB
#undef RRT2_MAXOBS
#undef RRT2_NU
#undef RRT2_EBD
#undef RRT2_HW
#undef RRT2_FCAP
#undef RRT2_WPS

#define RRT2_NS rppk2t
#define RRT2_TPB 64
#define RRT2_MAXOBS 56
#define RRT2_NU 44
#define RRT2_EBD 16
#define RRT2_HW 24
#define RRT2_FCAP 32
#define RRT2_WPS 4
#include "rrt_star_v2_body.inc"
#undef RRT2_NS
#undef RRT2_TPB
#undef RRT2_MAXOBS
#undef RRT2_NU
#undef RRT2_EBD
#undef RRT2_HW
#undef RRT2_FCAP
#undef RRT2_WPS
