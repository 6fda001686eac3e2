// Micro-benchmark (diagnostic, not part of the product): what a DEPENDENT load costs a wave while the other waves of
// its CU stream, in the occupancy shape of the C2 kernel (64-thread workgroups, 10 KB LDS each -> 16 per CU, 4096
// workgroups).  Every workgroup alternates a streaming phase (its own 200 KB, non-temporal 16-byte loads, D in
// flight per lane) with a latency phase of H dependent hops of one kind:
//   kind 0  vector load chain through a small (64 KB, L2-resident) table
//   kind 1  vector load chain through a 1 GiB table (HBM / Infinity Cache miss)
//   kind 2  scalar load chain (s_load_dword) through the small table
//   kind 3  scalar load chain with glc through the 1 GiB table
//   kind 4  LDS chain (reference point)
// Prints ns per hop, the per-wave streaming time for 200 KB and the chip-wide streaming rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o lat_ubench lat_ubench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ stream, int64_t per_block_words,
                                        const uint32_t* small_tab, uint32_t small_mask, const uint32_t* big_tab,
                                        uint32_t big_mask, int kind, int hops, int rounds, uint64_t* t_hop,
                                        uint64_t* t_stream, uint32_t* sink) {
  __shared__ uint32_t lds[2560];   // 10 KB: 16 workgroups per CU
  const int lane = threadIdx.x;
  for (int i = lane; i < 2560; i += 64) lds[i] = (i * 2654435761u) % 2560u;
  __syncthreads();
  const uint32_t* p = stream + (int64_t)blockIdx.x * per_block_words;
  const int nvec = (int)(per_block_words / 4);   // v4u elements of this block
  uint32_t acc = 0, idx = (blockIdx.x * 977u + 13u);
  uint64_t th = 0, ts = 0;
  // desynchronise the workgroups (as the planner's instances are)
  for (int r = 0; r < rounds; r++) {
    uint64_t a0 = __builtin_amdgcn_s_memrealtime();
    {
      v4u q[D];
      const v4u* pv = reinterpret_cast<const v4u*>(p);
#pragma unroll
      for (int u = 0; u < D; u++) q[u] = __builtin_nontemporal_load(pv + u * 64 + lane);
      for (int base = 0; base < nvec; base += 64 * D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
          const v4u cur = q[u];
          int nx = base + 64 * D + u * 64 + lane;
          nx = nx < nvec ? nx : lane;   // tail: harmless re-read
          q[u] = __builtin_nontemporal_load(pv + nx);
          acc += cur.x ^ cur.y ^ cur.z ^ cur.w;
        }
      }
#pragma unroll
      for (int u = 0; u < D; u++) acc += q[u].x;
    }
    uint64_t a1 = __builtin_amdgcn_s_memrealtime();
    if (kind == 0) {
      for (int h = 0; h < hops; h++) idx = small_tab[idx & small_mask];
    } else if (kind == 1) {
      for (int h = 0; h < hops; h++) idx = big_tab[idx & big_mask];
    } else if (kind == 2) {
      uint32_t s = __builtin_amdgcn_readfirstlane(idx);
      for (int h = 0; h < hops; h++) {
        uint32_t off = (s & small_mask) * 4u, v;
        asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(small_tab), "s"(off) : "memory");
        s = v;
      }
      idx = s;
    } else if (kind == 3) {
      uint32_t s = __builtin_amdgcn_readfirstlane(idx);
      for (int h = 0; h < hops; h++) {
        // byte offset up to 4 GiB: add to the base
        const uint32_t* q = big_tab + (s & big_mask);
        uint32_t v;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q) : "memory");
        s = v;
      }
      idx = s;
    } else {
      for (int h = 0; h < hops; h++) idx = lds[idx % 2560u];
    }
    uint64_t a2 = __builtin_amdgcn_s_memrealtime();
    asm volatile("" ::"v"(idx));
    ts += a1 - a0;
    th += a2 - a1;
  }
  if (lane == 0) {
    t_hop[blockIdx.x] = th;
    t_stream[blockIdx.x] = ts;
  }
  sink[blockIdx.x * 64 + lane] = acc + idx;
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main() {
  const int B = 4096;
  const int64_t per_block_bytes = 200 * 1024, per_block_words = per_block_bytes / 4;
  uint32_t *stream, *small_tab, *big_tab, *sink;
  uint64_t *t_hop, *t_stream;
  const uint32_t small_n = 16384, big_n = 1u << 28;   // 64 KB, 1 GiB
  CK(hipMalloc(&stream, (size_t)B * per_block_bytes));
  CK(hipMemset(stream, 1, (size_t)B * per_block_bytes));
  CK(hipMalloc(&small_tab, small_n * 4));
  CK(hipMalloc(&big_tab, (size_t)big_n * 4));
  CK(hipMalloc(&sink, B * 64 * 4));
  CK(hipMalloc(&t_hop, B * 8));
  CK(hipMalloc(&t_stream, B * 8));
  {
    std::vector<uint32_t> t(small_n);
    uint32_t s = 12345;
    for (uint32_t i = 0; i < small_n; i++) { s = s * 1664525u + 1013904223u; t[i] = s >> 8; }
    CK(hipMemcpy(small_tab, t.data(), small_n * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> b(big_n);
    for (uint32_t i = 0; i < big_n; i++) { s = s * 1664525u + 1013904223u; b[i] = s >> 3; }
    CK(hipMemcpy(big_tab, b.data(), (size_t)big_n * 4, hipMemcpyHostToDevice));
  }
  const char* kn[] = {"vector/L2-table", "vector/1GiB", "scalar/L2-table", "scalar-glc/1GiB", "LDS"};
  std::vector<uint64_t> hh(B), hs(B);
  printf("%-18s %5s %5s | %10s %14s %12s\n", "kind", "depth", "hops", "ns/hop", "us/200KB/wave", "chip TB/s");
  for (int depth : {4, 8, 12}) {
    for (int kind = 0; kind < 5; kind++) {
      for (int hops : {0, 24}) {
        if (hops == 0 && kind != 0) continue;
        const int rounds = 40;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
#define LAUNCH(D) hipLaunchKernelGGL(k<D>, dim3(B), dim3(64), 0, 0, stream, per_block_words, small_tab, small_n - 1, \
                                     big_tab, big_n - 1, kind, hops, rounds, t_hop, t_stream, sink)
        if (depth == 4) LAUNCH(4); else if (depth == 8) LAUNCH(8); else LAUNCH(12);
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        CK(hipMemcpy(hh.data(), t_hop, B * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), t_stream, B * 8, hipMemcpyDeviceToHost));
        double sh = 0, ss = 0;
        for (int i = 0; i < B; i++) { sh += hh[i]; ss += hs[i]; }
        const double ns_hop = hops ? sh * 10.0 / B / rounds / hops : 0.0;   // s_memrealtime: 100 MHz
        const double us_stream = ss * 10.0 / B / rounds / 1e3;
        const double tbs = (double)B * per_block_bytes * rounds / (ms * 1e-3) / 1e12;
        printf("%-18s %5d %5d | %10.0f %14.1f %12.2f   (kernel %.1f ms)\n", kn[kind], depth, hops, ns_hop, us_stream, tbs, ms);
        fflush(stdout);
      }
    }
  }
  return 0;
}
