// Micro-benchmark (diagnostic): shader cycles per call of the scalar arithmetic replicas on one wave, by the number of
// active lanes (each lane its own argument, so table lookups differ per lane).  Build twice to compare the table paths:
//   hipcc ... -DRPP_ROM_SGATHER_MAX=0  (every lookup a vector load)   |   default (scalar-path gather up to 16 lanes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "rpp_core.h"
#ifndef RPP_ROM_SGATHER_MAX
#define RPP_ROM_SGATHER_MAX 16   // the header's default (device side only)
#endif
__global__ void k(int op, int n, int lanes, double a0, double b0, double* out, long long* cyc) {
  double a = a0 + threadIdx.x * 0.37, b = b0, acc = 0.0;
  long long t0 = __builtin_amdgcn_s_memtime();
  if ((int)threadIdx.x < lanes) {
    for (int i = 0; i < n; i++) {
      double r;
      switch (op) {
        case 0: r = rpp::py_hypot(a, b); break;
        case 1: r = rpp::py_sq(a); break;
        case 2: r = rpp_glibc_sin(a); break;
        case 3: r = rpp_glibc_cos(a); break;
        case 4: r = rpp_glibc_atan2(a, b); break;
        case 5: r = rpp_glibc_acos(a * 0.01); break;
        default: { rpp::Edge e; rpp::steer(&e, 0.0, 0.0, a, b, 2.0, 0.25); r = e.ex; } break;
      }
      acc += r; a += r * 1e-9;   // dependent chain
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  if (hipMalloc(&out, 8 * 64) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
  const char* names[] = {"hypot", "sq(pow)", "sin", "cos", "atan2", "acos", "steer(8 steps)"};
  printf("RPP_ROM_SGATHER_MAX=%d\n", RPP_ROM_SGATHER_MAX);
  for (int lanes : {1, 4, 8, 16, 32, 64}) for (int op = 0; op < 7; op++) {
    int n = 2000; long long h = 0;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, op, n, lanes, 1.234, 0.777, out, cyc);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("lanes=%2d %-16s %8.1f cycles/call\n", lanes, names[op], (double)h / n);
  }
  return 0;
}
