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
// plan_div (common.h): the fp32-reciprocal division of the launch plan must equal x / c for 0 <= x < 2^22, 0 < c < 2^22
__global__ void kdiv(const int* xs, const int* cs, int* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = plan_div(xs[i], cs[i]);
}
static int check_plan_div() {
  const int n = 1 << 20;
  std::vector<int> xs(n), cs(n), o(n);
  srand(11);
  for (int i = 0; i < n; ++i) {
    const int c = 1 + (i % 3 == 0 ? rand() % 64 : i % 3 == 1 ? rand() % 4096 : rand() % (1 << 22));
    int x;
    switch (rand() % 4) {                                   // multiples of c and their neighbours are the hard cases
      case 0: x = (rand() % ((1 << 22) / c + 1)) * c; break;
      case 1: x = (rand() % ((1 << 22) / c + 1)) * c - 1; break;
      case 2: x = (rand() % ((1 << 22) / c + 1)) * c + 1; break;
      default: x = (int)(((long long)rand() * 4099 + rand()) % (1 << 22));
    }
    if (x < 0) x = 0;
    if (x >= (1 << 22)) x = (1 << 22) - 1;
    xs[i] = x; cs[i] = c;
  }
  int *dx, *dc, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, cs.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(kdiv, dim3(n / 256), dim3(256), 0, 0, dx, dc, dout, n);
  hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i)
    if (o[i] != xs[i] / cs[i]) { if (bad++ < 10) std::printf("plan_div(%d, %d) = %d, want %d\n", xs[i], cs[i], o[i], xs[i] / cs[i]); }
  std::printf(bad ? "plan_div: %d mismatches\n" : "plan_div ok (%d cases)\n", bad ? bad : n);
  return bad;
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
  bad += check_plan_div();
  return bad != 0;
}
