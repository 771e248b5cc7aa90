// flow_affine.hip -- optical-flow update after two affine warps.
//
// Restates extensions/flow_affine_transformation/flow_affine_transformation.cpp:63-83 per pixel
// (i = y, j = x), in IEEE fp32 with one rounding per operation, left-to-right:
//     x2 = round(M2[0]*j + M2[1]*i + M2[2]);  y2 = round(M2[3]*j + M2[4]*i + M2[5])
//     x1 = j + f.x;  y1 = i + f.y
//     x1 = round(M1[0]*x1 + M1[1]*y1 + M1[2])
//     y1 = round(M1[3]*x1 + M1[4]*y1 + M1[5])     <- uses the UPDATED x1 (.cpp:72-73), kept
//     clamp to [0, W-1] / [0, H-1];  out = (x1 - x2, y1 - y2)
// Compiled with FP contraction off (pragma below), so the result is bit-identical to the
// x86-64 build of the reference (g++ -O2 emits no FMA); rounding is an exact half-away-from-zero.
//
// The reference is a single-threaded scalar loop inside DataLoader workers.  Here it is a
// streaming kernel: one pixel (8 bytes in, 8 bytes out) per lane, one row band per block, so no
// integer division is needed and every access is a coalesced 512-byte wave transaction.
// Algorithmic HBM bytes: 16*H*W + 48.
#include "common.h"

// hipcc defaults to -ffp-contract=fast; contraction is switched off for this translation unit to
// keep one rounding per operation.  (HIP's __fmul_rn/__fadd_rn are inline header functions compiled
// BEFORE this pragma, so they still fuse -- the kernel uses plain operators instead.)
#pragma clang fp contract(off)

namespace rmnet {
namespace {

constexpr int kThreads = 256;

// std::round (half away from zero), exactly.  The device library's roundf evaluates
// trunc(x + copysign(0.5, x)), which rounds 3.4999998f up to 4 (the add is inexact); x - trunc(x)
// is always exact, so this form has no such case.
__device__ inline float round_half_away(float x) {
  float t = truncf(x);
  const float d = x - t;
  if (fabsf(d) >= 0.5f) t += copysignf(1.0f, x);
  return t;
}

__global__ __launch_bounds__(kThreads) void flow_affine_kernel(const float2* __restrict__ flow,
                                                               const float* __restrict__ m1,
                                                               const float* __restrict__ m2,
                                                               int H, int W,
                                                               float2* __restrict__ out) {
  const float a0 = m1[0], a1 = m1[1], a2 = m1[2], a3 = m1[3], a4 = m1[4], a5 = m1[5];
  const float b0 = m2[0], b1 = m2[1], b2 = m2[2], b3 = m2[3], b4 = m2[4], b5 = m2[5];
  const float fw = (float)W, fh = (float)H, fw1 = (float)(W - 1), fh1 = (float)(H - 1);
  const int x = blockIdx.x * kThreads + threadIdx.x;
  if (x >= W) return;
  const float fj = (float)x;
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const float fi = (float)y;
    const size_t idx = (size_t)y * W + x;
    const float2 f = flow[idx];
    // plain operators: they, unlike the header-defined __fmul_rn/__fadd_rn, obey the pragma above
    float x2 = round_half_away(b0 * fj + b1 * fi + b2);
    float y2 = round_half_away(b3 * fj + b4 * fi + b5);
    float x1 = fj + f.x;
    float y1 = fi + f.y;
    x1 = round_half_away(a0 * x1 + a1 * y1 + a2);
    y1 = round_half_away(a3 * x1 + a4 * y1 + a5);
    x1 = x1 < 0.0f ? 0.0f : (x1 >= fw ? fw1 : x1);
    y1 = y1 < 0.0f ? 0.0f : (y1 >= fh ? fh1 : y1);
    x2 = x2 < 0.0f ? 0.0f : (x2 >= fw ? fw1 : x2);
    y2 = y2 < 0.0f ? 0.0f : (y2 >= fh ? fh1 : y2);
    out[idx] = make_float2(x1 - x2, y1 - y2);
  }
}

}  // namespace

int launch_flow_affine(const float* flow, const float* m1, const float* m2, int H, int W,
                       float* out, hipStream_t st) {
  if (!flow || !m1 || !m2 || !out || H <= 0 || W <= 0) return RMNET_E_INVALID_ARG;
  if (H >= (1 << 24) || W >= (1 << 24)) return RMNET_E_UNSUPPORTED;  // int -> float must be exact
  dim3 g((W + kThreads - 1) / kThreads, H < 4096 ? H : 4096);
  hipLaunchKernelGGL(flow_affine_kernel, g, dim3(kThreads), 0, st,
                     reinterpret_cast<const float2*>(flow), m1, m2, H, W,
                     reinterpret_cast<float2*>(out));
  return check_launch();
}

}  // namespace rmnet
