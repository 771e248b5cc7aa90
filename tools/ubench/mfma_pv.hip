// PV-shaped MFMA loop calibration: NDT A-fragments x 4 B-fragment pairs x 3 terms into NDT*4 accumulators,
// operands in distinct registers (as in bk_main), optional LDS reads of the B fragments per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NDT, bool LDSB>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int lane = threadIdx.x & 63;
  half8 vh[NDT], vl[NDT];
  for (int d = 0; d < NDT; ++d) for (int i = 0; i < 8; ++i) { vh[d][i] = (_Float16)(lane * 0.01f + d + i); vl[d][i] = (_Float16)(0.001f * (d + i)); }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  f32x4 acc[NDT][4];
  for (int d = 0; d < NDT; ++d) for (int i = 0; i < 4; ++i) acc[d][i] = f32x4{0, 0, 0, 0};
  half8 bh[4], bl[4];
  for (int it = 0; it < 4; ++it) for (int i = 0; i < 8; ++i) { bh[it][i] = (_Float16)(0.5f + it); bl[it][i] = (_Float16)(0.01f * it); }
  long long t0 = __builtin_readcyclecounter();
  for (int n = 0; n < iters; ++n) {
    if (LDSB) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        bh[it] = *reinterpret_cast<const half8*>(lds + it * 2048 + lane * 16);
        bl[it] = *reinterpret_cast<const half8*>(lds + it * 2048 + 1024 + lane * 16);
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
#pragma unroll
      for (int d = 0; d < NDT; ++d) acc[d][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[d], bh[it], acc[d][it], 0, 0, 0);
#pragma unroll
      for (int d = 0; d < NDT; ++d) acc[d][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[d], bl[it], acc[d][it], 0, 0, 0);
#pragma unroll
      for (int d = 0; d < NDT; ++d) acc[d][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[d], bh[it], acc[d][it], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int d = 0; d < NDT; ++d) for (int i = 0; i < 4; ++i) s += acc[d][i][0];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = s; out[blockIdx.x * 2 + 1] = (float)(t1 - t0) / (iters * 12.0f * NDT); }
}

template <int NDT, bool LDSB>
void run(float* out, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NDT, LDSB>), dim3(256), dim3(threads), 0, 0, out, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NDT, LDSB>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms, h[2]; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
  const double n_mfma_wave = iters * 12.0 * NDT;
  printf("waves/SIMD %d NDT=%d ldsB=%d: %.1f ticks/MFMA/wave, %.2f ns/MFMA/wave wall, %.0f TF/s\n", threads / 256, NDT, (int)LDSB, h[1],
         ms * 1e6 / n_mfma_wave, 256.0 * (threads / 64) * n_mfma_wave * 16384 / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 8);
  run<5, false>(out, 256); run<5, false>(out, 512); run<5, true>(out, 256); run<5, true>(out, 512); run<3, true>(out, 512);
  return 0;
}
