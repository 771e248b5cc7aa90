// Checks the DPP / permlane wave reductions and scan of common.h against the __shfl_* forms on random inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/dpp_check.hip -o tools/ubench/dpp_check && tools/ubench/dpp_check
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../rmnet_amd/csrc/common.h"
using namespace rmnet;
__global__ void k(const int* in, int* out) {
  const int l = threadIdx.x, v = in[blockIdx.x * 64 + l];
  int s = v, m = v, sc = v;
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); m = max(m, __shfl_xor(m, o)); }
  for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(sc, d); if (l >= d) sc += u; }
  int* o = out + (blockIdx.x * 64 + l) * 6;
  o[0] = s; o[1] = wave_sum_fast(v); o[2] = m; o[3] = wave_max_fast(v); o[4] = sc; o[5] = wave_scan_incl_fast(v);
}
int main() {
  const int nb = 256;
  std::vector<int> h(nb * 64);
  srand(3);
  for (auto& x : h) x = rand() % 20001 - 10000;
  for (int i = 0; i < 64; ++i) h[i] = i == 5 ? 7 : 0;          // a sparse block
  int *din, *dout;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dout, h.size() * 24);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, din, dout);
  std::vector<int> o(h.size() * 6);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (size_t i = 0; i < h.size(); ++i)
    for (int j = 0; j < 3; ++j)
      if (o[i * 6 + 2 * j] != o[i * 6 + 2 * j + 1]) { if (bad++ < 10) std::printf("lane %zu kind %d: shfl %d dpp %d\n", i % 64, j, o[i * 6 + 2 * j], o[i * 6 + 2 * j + 1]); }
  std::printf(bad ? "dpp_check: %d mismatches\n" : "dpp_check ok (%d)\n", bad ? bad : (int)h.size());
  return bad != 0;
}
