// epilogue.hip -- per-channel affine + residual + ReLU in ONE streaming pass (gfx950).
//
// The convolutions of RMNet's encoders/decoder stay on MIOpen, but the
// elementwise glue between them is this repo's: in eval mode the reference runs, per convolution,
// BatchNorm -> (+ skip) -> ReLU as up to three more full passes over the activation
// (torchvision Bottleneck under models/rmnet.py:66-80; ResBlock, models/rmnet.py:24-48, adds the
// conv bias, the ReLU and the skip the same way).  All of these are
//     out[n, c, :] = act( x[n, c, :] * scale[c] + shift[c]  +  (res[n, c, :] * rscale[c] + rshift[c]) )
// with scale/shift = the folded BatchNorm statistics or (1, conv bias), so one HBM-bound kernel
// replaces them: 8 B (or 12 B with a residual) of traffic per element instead of 16-32 B.
//   grid = (chunks of a plane, planes n*C + c): scale/shift are wave-uniform scalars, the plane is
//   streamed with 16-byte loads/stores, 4 in flight per lane.  In place (out == x or out == res) is fine.
#include "common.h"

#pragma clang fp contract(off)   // x*scale + shift rounds like the two torch ops it replaces

namespace rmnet {
namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 4;

template <bool VEC4, bool RES>
__global__ __launch_bounds__(kThreads) void channel_affine(const float* __restrict__ x,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* res,
                                                           const float* __restrict__ rscale,
                                                           const float* __restrict__ rshift,
                                                           int relu, int C, long long planes,
                                                           long long HW, float* out) {
  for (long long plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const int c = (int)(plane % C);
    const float sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
    const float rs = RES && rscale ? rscale[c] : 1.0f, rh = RES && rshift ? rshift[c] : 0.0f;
    const float lo = relu ? 0.0f : -INFINITY;
    const size_t base = (size_t)plane * HW;
    auto f = [&](float v, float r) {
      float y = v * sc + sh;
      if (RES) y = y + (r * rs + rh);
      return y < lo ? lo : y;          // NaN stays NaN (torch.relu semantics)
    };
    if (VEC4) {
      const long long n4 = HW >> 2;
      const float4* x4 = reinterpret_cast<const float4*>(x + base);
      const float4* r4 = reinterpret_cast<const float4*>(res + base);
      float4* o4 = reinterpret_cast<float4*>(out + base);
      for (long long i0 = (long long)blockIdx.x * kThreads * kUnroll; i0 < n4;
           i0 += (long long)gridDim.x * kThreads * kUnroll) {
        float4 v[kUnroll], r[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const long long i = i0 + u * kThreads + threadIdx.x;
          v[u] = i < n4 ? x4[i] : float4{0.f, 0.f, 0.f, 0.f};
          if (RES) r[u] = i < n4 ? r4[i] : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const long long i = i0 + u * kThreads + threadIdx.x;
          if (i < n4) {
            float4 y;
            y.x = f(v[u].x, RES ? r[u].x : 0.f); y.y = f(v[u].y, RES ? r[u].y : 0.f);
            y.z = f(v[u].z, RES ? r[u].z : 0.f); y.w = f(v[u].w, RES ? r[u].w : 0.f);
            o4[i] = y;
          }
        }
      }
    } else {
      for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < HW;
           i += (long long)gridDim.x * kThreads)
        out[base + i] = f(x[base + i], RES ? res[base + i] : 0.f);
    }
  }
}

}  // namespace

int launch_channel_affine(const float* x, const float* scale, const float* shift, const float* res,
                          const float* rscale, const float* rshift, int relu, long long N, int C,
                          long long HW, float* out, hipStream_t st) {
  if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return RMNET_E_INVALID_ARG;
  if (!res && (rscale || rshift)) return RMNET_E_INVALID_ARG;
  const long long planes = N * C;
  const bool vec = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                                      reinterpret_cast<uintptr_t>(res)) & 15) == 0;
  const long long per_block = (long long)kThreads * (vec ? 4 * kUnroll : 1);
  long long chunks = (HW + per_block - 1) / per_block;
  if (chunks > 64) chunks = 64;
  // enough workgroups to fill the chip even for few planes, never more than the grid limits
  const dim3 grid((unsigned)chunks, (unsigned)(planes < 65535 ? planes : 65535));
  const dim3 block(kThreads);
#define RMNET_CA(V, R) hipLaunchKernelGGL((channel_affine<V, R>), grid, block, 0, st, x, scale, shift, res, rscale, rshift, relu, C, planes, HW, out)
  if (vec) { if (res) RMNET_CA(true, true); else RMNET_CA(true, false); }
  else     { if (res) RMNET_CA(false, true); else RMNET_CA(false, false); }
#undef RMNET_CA
  return check_launch();
}

}  // namespace rmnet
