// Micro-benchmark (diagnostic): shader cycles per call of the scalar arithmetic replicas on one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "rpp_core.h"
__global__ void k(int op, int n, double a0, double b0, double* out, long long* cyc) {
  double a = a0 + threadIdx.x * 1e-3, b = b0, acc = 0.0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i++) {
    double r;
    switch (op) {
      case 0: r = rpp::py_hypot(a, b); break;
      case 1: r = rpp::py_sq(a); break;
      case 2: r = rpp_glibc_sin(a); break;
      case 3: r = rpp_glibc_cos(a); break;
      case 4: r = rpp_glibc_atan2(a, b); break;
      default: { rpp::Edge e; rpp::steer(&e, 0.0, 0.0, a, b, 2.0, 0.25); r = e.ex; } break;
    }
    acc += r; a += r * 1e-9;   // dependent chain
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8);
  const char* names[] = {"hypot", "sq(pow)", "sin", "cos", "atan2", "steer(8 steps)"};
  for (int lanes : {1, 64}) for (int op = 0; op < 6; op++) {
    int n = 2000; long long h;
    hipLaunchKernelGGL(k, dim3(1), dim3(lanes), 0, 0, op, n, 1.234, 0.777, out, cyc);
    hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("lanes=%2d %-16s %8.1f cycles/call\n", lanes, names[op], (double)h / n);
  }
  return 0;
}
