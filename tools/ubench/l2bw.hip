// Micro-benchmark: per-CU read bandwidth from an L2-resident buffer on gfx950, plain 16-byte loads to
// VGPRs (with D loads in flight per lane) vs LDS-DMA.  Build: hipcc --offload-arch=gfx950 -O3 l2bw.hip -o l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void plain(const f4* __restrict__ src, size_t n_f4, int iters, float* sink) {
  const int tid = threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  const size_t per_iter = 512 * D;
  size_t base = ((size_t)blockIdx.x * 7919 * per_iter) % n_f4;
  for (int it = 0; it < iters; ++it) {
    f4 v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = src[(base + (size_t)d * 512 + tid) % n_f4];
#pragma unroll
    for (int d = 0; d < D; ++d) acc += v[d];
    base = (base + per_iter) % n_f4;
  }
  if (acc.x == 12345.f) sink[0] = acc.y + acc.z + acc.w;
}

__global__ __launch_bounds__(512) void dma(const f4* __restrict__ src, size_t n_f4, int iters, float* sink, int D) {
  __shared__ __attribute__((aligned(16))) char lds[8 * 1024 * 8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t per_iter = 512 * (size_t)D;
  size_t base = ((size_t)blockIdx.x * 7919 * per_iter) % n_f4;
  float s = 0;
  for (int it = 0; it < iters; ++it) {
    for (int d = 0; d < D; ++d) {
      const f4* g = src + (base + (size_t)d * 512 + tid) % n_f4;
      __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(lds + (wave * 8 + (d & 7)) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s += *(float*)(lds + (wave * 8) * 1024 + lane * 16);
    base = (base + per_iter) % n_f4;
  }
  if (s == 12345.f) sink[0] = s;
}

int run_size(size_t mb) {
  const size_t bytes = mb << 20;
  const size_t n_f4 = bytes / 16;
  printf("---- footprint %zu MB (every WG streams the same region from a different offset)\n", mb);
  f4* src; float* sink;
  hipMalloc(&src, bytes); hipMalloc(&sink, 64); hipMemset(src, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, int D, int grid) {
    const int iters = 4096 / D;
    launch(grid, iters); hipDeviceSynchronize();
    hipEventRecord(e0); launch(grid, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)grid * iters * 512.0 * D * 16.0;
    printf("%-10s D=%2d grid=%3d : %8.1f GB/s total, %6.1f GB/s per WG (%.1f B/clk at 2.1 GHz)\n", name, D, grid,
           total / ms / 1e6, total / ms / 1e6 / grid, total / ms / 1e6 / grid / 2.1);
  };
  for (int grid : {256}) {
    run("plain", [&](int g, int it) { hipLaunchKernelGGL(plain<8>, dim3(g), dim3(512), 0, 0, src, n_f4, it, sink); }, 8, grid);
    run("plain", [&](int g, int it) { hipLaunchKernelGGL(plain<16>, dim3(g), dim3(512), 0, 0, src, n_f4, it, sink); }, 16, grid);
  }
  hipFree(src); hipFree(sink);
  return 0;
}

int main() {
  for (size_t mb : {2, 4, 8, 16, 32, 128, 512}) run_size(mb);
  return 0;
}
