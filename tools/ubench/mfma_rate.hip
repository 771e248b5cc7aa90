// MFMA issue-rate calibration on gfx950: v_mfma_f32_16x16x32_f16 with NACC independent accumulators,
// 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(float* out, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = s; out[blockIdx.x * 2 + 1] = (float)(t1 - t0) / (iters * 3.0f * NACC); }
}

template <int NACC>
void run(float* out, int threads) {
  float h[8];
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  const double flops = 256.0 * (threads / 64) * iters * 3.0 * NACC * 16384.0;
  printf("waves/SIMD %d NACC=%d: %.1f memtime-ticks/MFMA/wave, %.0f TF/s wall (%.2f ms)\n", threads / 256, NACC, h[1], flops / ms / 1e9, ms);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 8);
  run<1>(out, 256); run<4>(out, 256); run<8>(out, 256); run<1>(out, 512); run<4>(out, 512); run<8>(out, 512);
  return 0;
  float h[8];
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, 2000); hipDeviceSynchronize(); hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("threads/WG %d (waves/SIMD %d): NACC=1 (dependent chain) %.1f cycles/MFMA/wave\n", threads, threads / 256, h[1]);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, 2000); hipDeviceSynchronize(); hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("threads/WG %d: NACC=2 %.1f\n", threads, h[1]);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, 2000); hipDeviceSynchronize(); hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("threads/WG %d: NACC=4 %.1f\n", threads, h[1]);
    hipLaunchKernelGGL(k<8>, dim3(256), dim3(threads), 0, 0, out, 2000); hipDeviceSynchronize(); hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("threads/WG %d: NACC=8 %.1f\n", threads, h[1]);
  }
  return 0;
}
