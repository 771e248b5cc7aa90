// Sustained MFMA throughput and shader clock vs operand data and kernel duration (power / DVFS):
// all 256 CUs, W waves per SIMD, pure v_mfma loops on register operands, zero or random fp16 data.
// hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(768) void k(const half8* __restrict__ src, long long* out, int iters, int nops) {
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 1024 * i) & 4095]; b[i] = src[(threadIdx.x + 1024 * i + 517) & 4095]; }
  const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  if (SHAPE == 0) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 1) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  }
  const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = c1 - c0; out[blockIdx.x * 4 + 1] = r1 - r0; out[blockIdx.x * 4 + 2] = (long long)s; }
}

int main() {
  half8* src; long long* out;
  hipMalloc(&src, 4096 * 16); hipMalloc(&out, 256 * 32);
  _Float16* h = (_Float16*)malloc(4096 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int data = 0; data < 3; ++data) {
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i)
      h[i] = data == 0 ? (_Float16)0.0f : data == 1 ? (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f)
                                                   : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f * ((i & 1) ? 1.0f : 0.0005f));
    hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape)
      for (int threads : {256, 512, 768})
        for (int iters : {1000, 8000, 64000}) {
          const int per_it = shape == 0 ? 8 : 4;
          const double fl = shape == 0 ? 16384.0 : 32768.0;
          for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, src, out, iters, 0);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, src, out, iters, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
          }
          float ms; hipEventElapsedTime(&ms, e0, e1);
          long long o[4]; hipMemcpy(o, out, 32, hipMemcpyDeviceToHost);
          const double flops = 256.0 * (threads / 64) * (double)iters * per_it * fl;
          printf("data=%s shape=%s waves/SIMD=%d iters=%6d: %8.1f us  %7.0f TF/s  clock %.3f GHz  %.1f cyc/MFMA/SIMD\n",
                 data == 0 ? "zero  " : data == 1 ? "random" : "hi/lo ", shape == 0 ? "16x16x32" : "32x32x16", threads / 256, iters, ms * 1e3,
                 flops / ms / 1e9, o[0] / (o[1] * 10.0), (double)o[0] / ((double)iters * per_it * (threads / 256)));
        }
  }
  return 0;
}
