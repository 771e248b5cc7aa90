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
__global__ __launch_bounds__(kThreads) void channel_affine(const float* x,   // (may alias out: no restrict)
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
    const float lo = relu == 1 ? 0.0f : -INFINITY;
    const size_t base = (size_t)plane * HW;
    auto f = [&](float v, float r) {
      float y = v * sc + sh;
      if (RES) y = y + (r * rs + rh);
      if (relu == 2) return y > 0.0f ? y : y * 0.1f;   // LeakyReLU(0.1) of TinyFlowNet
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

// ---------------------------------------------------------------------------------------------
// x2 bilinear upsample (align_corners = False) fused with the skip add of Refine
// (models/rmnet.py:117-119: s + F.interpolate(pm, scale_factor=2, mode='bilinear')).
// torch's generic upsample kernel spends ~480 us on the [4,256,60,108] -> [4,256,120,216] map of the
// 480p decoder (0.2 TB/s) and the add is one more pass; here one thread produces 4 consecutive
// outputs of a row from 2 x 4 source values (weights are exactly 0.25 / 0.75, or 0 / 1 at the
// border clamp of area_pixel_compute_source_index), adds the skip and stores 16 bytes.
//   src = 0.5 * (dst + 0.5) - 0.5, clamped at 0;  i0 = floor(src);  i1 = i0 + (i0 < n - 1);
//   l1 = src - i0;  l0 = 1 - l1;  out = l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11)
// (aten/src/ATen/native/cuda/UpSampleBilinear2d.cu: upsample_bilinear2d_out_frame).
struct Tap { int i0, i1; float l0, l1; };
__device__ inline Tap tap2x(int d, int n) {
  float src = 0.5f * ((float)d + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  Tap t;
  t.i0 = (int)src;
  t.i1 = t.i0 + (t.i0 < n - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.0f - t.l1;
  return t;
}

template <bool VEC4, bool ADD>
__global__ __launch_bounds__(kThreads) void upsample2x_add(const float* __restrict__ x,
                                                           const float* skip, float* out, int h,
                                                           int w, long long planes) {
  const int H = 2 * h, W = 2 * w;
  const int per_row = VEC4 ? W >> 2 : W;
  const long long items = (long long)H * per_row;
  for (long long plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const float* xp = x + (size_t)plane * h * w;
    const size_t ob = (size_t)plane * H * W;
    for (long long it = (long long)blockIdx.x * kThreads + threadIdx.x; it < items;
         it += (long long)gridDim.x * kThreads) {
      const int y = (int)(it / per_row), xq = (int)(it - (long long)y * per_row);
      const Tap ty = tap2x(y, h);
      const float* r0 = xp + (size_t)ty.i0 * w;
      const float* r1 = xp + (size_t)ty.i1 * w;
      if (VEC4) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const Tap tx = tap2x(4 * xq + e, w);
          o[e] = ty.l0 * (tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1]) + ty.l1 * (tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1]);
        }
        const size_t oi = ob + (size_t)y * W + 4 * xq;
        if (ADD) {
          const float4 s = *reinterpret_cast<const float4*>(skip + oi);
          o[0] = s.x + o[0]; o[1] = s.y + o[1]; o[2] = s.z + o[2]; o[3] = s.w + o[3];
        }
        *reinterpret_cast<float4*>(out + oi) = float4{o[0], o[1], o[2], o[3]};
      } else {
        const Tap tx = tap2x(xq, w);
        float o = ty.l0 * (tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1]) + ty.l1 * (tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1]);
        const size_t oi = ob + (size_t)y * W + xq;
        if (ADD) o = skip[oi] + o;
        out[oi] = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Decoder tail: 2-class soft-max -> foreground probability -> soft aggregation
// (models/rmnet.py:289-302: background = prod(1 - p_o), clamp to [1e-7, 1 - 1e-7], logit =
// log(em / (1 - em))) -> un-pad (:375-380) -> optional soft-max over the K channels (:450) in ONE pass.
// The module graph spends ~25 small launches per frame on [B,K,H,W]-sized tensors for this.
//   dec [n_tot, 2, Hp, Wp] decoder logits of the objects in flight, clip b owning objects
//   [obj_begin[b], obj_begin[b+1]);  outputs [B, K, H, W] (channel 0 = background, 1..n = objects,
//   the rest absent: em = 0 -> clamp -> logit = log(1e-7 / (1 - 1e-7))).
__device__ inline float fg_prob(const float* __restrict__ dec, size_t plane, size_t pix) {
  const float z0 = dec[pix], z1 = dec[plane + pix];
  const float m = fmaxf(z0, z1);
  const float e0 = expf(z0 - m), e1 = expf(z1 - m);
  return e1 / (e0 + e1);
}
__device__ inline float em_logit(float em) {
  em = fminf(fmaxf(em, 1e-7f), 1.0f - 1e-7f);
  return logf(em / (1.0f - em));
}

__global__ __launch_bounds__(kThreads) void soft_aggregate(const float* __restrict__ dec,
                                                           const int32_t* __restrict__ obj_begin,
                                                           int K, int Hp, int Wp, int pad_l,
                                                           int pad_t, int H, int W,
                                                           float* __restrict__ logit,
                                                           float* __restrict__ prob) {
  const int b = blockIdx.y;
  const int o0 = obj_begin[b], n = min(obj_begin[b + 1] - o0, K - 1);
  const size_t plane = (size_t)Hp * Wp, oplane = (size_t)H * W;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < (long long)oplane;
       i += (long long)gridDim.x * kThreads) {
    const int y = (int)(i / W), x = (int)(i - (long long)y * W);
    const size_t pix = (size_t)(y + pad_t) * Wp + (x + pad_l);
    const float* d = dec + (size_t)o0 * 2 * plane;
    float* lo = logit + (size_t)b * K * oplane + i;
    float bg = 1.0f, mx = 0.0f;
    for (int o = 0; o < n; ++o) {
      const float p = fg_prob(d + (size_t)o * 2 * plane, plane, pix);
      bg = bg * (1.0f - p);
      const float l = em_logit(p);
      lo[(size_t)(o + 1) * oplane] = l;
      mx = o == 0 ? l : fmaxf(mx, l);
    }
    const float l0 = em_logit(bg), labs = em_logit(0.0f);
    lo[0] = l0;
    for (int k = n + 1; k < K; ++k) lo[(size_t)k * oplane] = labs;
    if (prob) {   // soft-max over the K channels (models/rmnet.py:450)
      mx = n > 0 ? fmaxf(mx, l0) : l0;
      if (n + 1 < K) mx = fmaxf(mx, labs);
      float sum = expf(l0 - mx);
      const float eabs = expf(labs - mx);
      for (int k = n + 1; k < K; ++k) sum += eabs;
      for (int o = 0; o < n; ++o) sum += expf(lo[(size_t)(o + 1) * oplane] - mx);
      float* po = prob + (size_t)b * K * oplane + i;
      po[0] = expf(l0 - mx) / sum;
      for (int o = 0; o < n; ++o) po[(size_t)(o + 1) * oplane] = expf(lo[(size_t)(o + 1) * oplane] - mx) / sum;
      for (int k = n + 1; k < K; ++k) po[(size_t)k * oplane] = eabs / sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ResNet stem tail: relu(x * scale[c] + shift[c]) followed by MaxPool2d(3, stride 2, padding 1)
// (torchvision resnet50's bn1 / relu / maxpool under models/rmnet.py:74-76, 98-100) in one pass: the
// full-resolution activation (212 MB per encoder at eight 480p clips) is read once and never written.
// Padding counts as -inf like torch's max_pool2d; a NaN in the window gives NaN.
template <bool VEC4>
__global__ __launch_bounds__(kThreads) void affine_relu_maxpool(const float* __restrict__ x,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                int C, long long planes, int H, int W,
                                                                int Ho, int Wo, float* __restrict__ out) {
  const int per_row = VEC4 ? Wo >> 2 : Wo;          // VEC4: W = 2 Wo, Wo % 4 == 0, 16-byte aligned rows
  const int items = Ho * per_row;
  for (long long plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const int c = (int)(plane % C);
    const float sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
    const float* xp = x + (size_t)plane * H * W;
    float* op = out + (size_t)plane * Ho * Wo;
    auto act = [&](float v, bool& nan) {
      v = v * sc + sh;
      nan = nan || v != v;
      return v < 0.0f ? 0.0f : v;
    };
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < items; i += gridDim.x * kThreads) {
      const int yo = i / per_row, xq = i - yo * per_row;
      if (VEC4) {
        // 4 outputs from the 9 input columns 8 xq - 1 .. 8 xq + 7 of 3 rows: two 16-byte loads and
        // one scalar per row
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        bool nan[4] = {false, false, false, false};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
          const int y = 2 * yo + dy;
          if (y < 0 || y >= H) continue;
          const float* row = xp + (size_t)y * W + 8 * xq;
          const float4 a = *reinterpret_cast<const float4*>(row);
          const float4 bq = *reinterpret_cast<const float4*>(row + 4);
          const bool has_left = xq > 0;
          const float left = has_left ? row[-1] : 0.0f;
          const float v[9] = {left, a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              if (j == 0 && dx == 0 && !has_left) continue;
              const float t = act(v[2 * j + dx], nan[j]);
              best[j] = t > best[j] ? t : best[j];
            }
        }
        float4 o;
        o.x = nan[0] ? __builtin_nanf("") : best[0]; o.y = nan[1] ? __builtin_nanf("") : best[1];
        o.z = nan[2] ? __builtin_nanf("") : best[2]; o.w = nan[3] ? __builtin_nanf("") : best[3];
        *reinterpret_cast<float4*>(op + (size_t)yo * Wo + 4 * xq) = o;
      } else {
        float best = -INFINITY;
        bool nan = false;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
          const int y = 2 * yo + dy;
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int xx = 2 * xq + dx;
            if (y >= 0 && y < H && xx >= 0 && xx < W) {
              const float t = act(xp[(size_t)y * W + xx], nan);
              best = t > best ? t : best;
            }
          }
        }
        op[(size_t)yo * Wo + xq] = nan ? __builtin_nanf("") : best;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// [r6] The same three passes for CHANNELS-LAST activations ([N, H, W, C] in memory: what MIOpen's NHWC implicit-GEMM and CK
// convolutions produce and consume without the batched_transpose kernels that wrap them on NCHW tensors -- 5.6 % of a step, and the
// Winograd / rocBLAS kernels of the NCHW path are no faster than their NHWC counterparts; profiles/r06_conv_layout.md).  The channel is
// the innermost index: a thread owns 4 consecutive channels of a pixel (C % 4 == 0 for every layer of both networks), scale / shift
// come as float4 from the L1, rows are streamed with 16-byte accesses, 4 in flight per lane.  Same arithmetic, same rounding.
template <bool RES, bool FIXED>
__global__ __launch_bounds__(kThreads) void channel_affine_nhwc(const float* x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* res,
                                                                const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                                int relu, int C4, long long n4, float* out) {
  const float lo = relu == 1 ? 0.0f : -INFINITY;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* r4 = reinterpret_cast<const float4*>(res);
  float4* o4 = reinterpret_cast<float4*>(out);
  const float4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  auto f = [&](float v, float sc, float sh, float r, float rs, float rh) {
    float y = v * sc + sh;
    if (RES) y = y + (r * rs + rh);
    if (relu == 2) return y > 0.0f ? y : y * 0.1f;
    return y < lo ? lo : y;
  };
  auto params = [&](int c, float4& sc, float4& sh, float4& rs, float4& rh) {
    sc = scale ? reinterpret_cast<const float4*>(scale)[c] : one;
    sh = shift ? reinterpret_cast<const float4*>(shift)[c] : zero;
    rs = RES && rscale ? reinterpret_cast<const float4*>(rscale)[c] : one;
    rh = RES && rshift ? reinterpret_cast<const float4*>(rshift)[c] : zero;
  };
  // FIXED: C4 divides the workgroup size, so a thread's float4 index i = i0 + u * kThreads + tid (i0 a multiple of kThreads * kUnroll) has the
  // same channel group i % C4 = tid % C4 in EVERY step: its four parameter vectors are loaded once.  (Loaded per element they were four more
  // 16-byte loads beside the two of the data: the NHWC kernel ran at half the NCHW kernel's rate, profiles/r06_conv_layout.md.)
  float4 sc, sh, rs, rh;
  if (FIXED) params((int)(threadIdx.x % C4), sc, sh, rs, rh);
  for (long long i0 = (long long)blockIdx.x * kThreads * kUnroll; i0 < n4; i0 += (long long)gridDim.x * kThreads * kUnroll) {
    float4 v[kUnroll], r[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const long long i = i0 + u * kThreads + threadIdx.x;
      v[u] = i < n4 ? x4[i] : zero;
      if (RES) r[u] = i < n4 ? r4[i] : zero;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const long long i = i0 + u * kThreads + threadIdx.x;
      if (i < n4) {
        if (!FIXED) params((int)(i % C4), sc, sh, rs, rh);
        float4 y;
        y.x = f(v[u].x, sc.x, sh.x, RES ? r[u].x : 0.f, rs.x, rh.x); y.y = f(v[u].y, sc.y, sh.y, RES ? r[u].y : 0.f, rs.y, rh.y);
        y.z = f(v[u].z, sc.z, sh.z, RES ? r[u].z : 0.f, rs.z, rh.z); y.w = f(v[u].w, sc.w, sh.w, RES ? r[u].w : 0.f, rs.w, rh.w);
        o4[i] = y;
      }
    }
  }
}

// x [N, h, w, C] -> out [N, 2h, 2w, C] = skip + bilinear x2 (align_corners = False); one thread = 4 channels of one output pixel
template <bool ADD>
__global__ __launch_bounds__(kThreads) void upsample2x_add_nhwc(const float* __restrict__ x, const float* skip, float* out, int h, int w,
                                                                int C4, long long items) {
  const int H = 2 * h, W = 2 * w;
  for (long long it = (long long)blockIdx.x * kThreads + threadIdx.x; it < items; it += (long long)gridDim.x * kThreads) {
    const int c = (int)(it % C4);
    long long p = it / C4;
    const int xo = (int)(p % W); p /= W;
    const int yo = (int)(p % H);
    const long long n = p / H;
    const Tap ty = tap2x(yo, h), tx = tap2x(xo, w);
    const float4* xb = reinterpret_cast<const float4*>(x) + (size_t)n * h * w * C4 + c;
    const float4 v00 = xb[((size_t)ty.i0 * w + tx.i0) * C4], v01 = xb[((size_t)ty.i0 * w + tx.i1) * C4];
    const float4 v10 = xb[((size_t)ty.i1 * w + tx.i0) * C4], v11 = xb[((size_t)ty.i1 * w + tx.i1) * C4];
    auto g = [&](float a00, float a01, float a10, float a11) {
      return ty.l0 * (tx.l0 * a00 + tx.l1 * a01) + ty.l1 * (tx.l0 * a10 + tx.l1 * a11);
    };
    float4 o = {g(v00.x, v01.x, v10.x, v11.x), g(v00.y, v01.y, v10.y, v11.y), g(v00.z, v01.z, v10.z, v11.z), g(v00.w, v01.w, v10.w, v11.w)};
    if (ADD) {
      const float4 sk = reinterpret_cast<const float4*>(skip)[it];
      o.x = sk.x + o.x; o.y = sk.y + o.y; o.z = sk.z + o.z; o.w = sk.w + o.w;
    }
    reinterpret_cast<float4*>(out)[it] = o;
  }
}

// x [N, H, W, C] -> out [N, Ho, Wo, C] = MaxPool2d(3, 2, 1) of relu(x * scale + shift); one thread = 4 channels of one output pixel
__global__ __launch_bounds__(kThreads) void affine_relu_maxpool_nhwc(const float* __restrict__ x, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, int C4, int H, int W, int Ho, int Wo,
                                                                     long long items, float* __restrict__ out) {
  for (long long it = (long long)blockIdx.x * kThreads + threadIdx.x; it < items; it += (long long)gridDim.x * kThreads) {
    const int c = (int)(it % C4);
    long long p = it / C4;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const long long n = p / Ho;
    const float4 sc = scale ? reinterpret_cast<const float4*>(scale)[c] : float4{1.f, 1.f, 1.f, 1.f};
    const float4 sh = shift ? reinterpret_cast<const float4*>(shift)[c] : float4{0.f, 0.f, 0.f, 0.f};
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    bool nan[4] = {false, false, false, false};
    const float4* xb = reinterpret_cast<const float4*>(x) + (size_t)n * H * W * C4 + c;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int y = 2 * yo + dy;
      if (y < 0 || y >= H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = 2 * xo + dx;
        if (xx < 0 || xx >= W) continue;
        const float4 v = xb[((size_t)y * W + xx) * C4];
        const float t[4] = {v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          nan[e] = nan[e] || t[e] != t[e];
          const float a = t[e] < 0.0f ? 0.0f : t[e];
          best[e] = a > best[e] ? a : best[e];
        }
      }
    }
    const float qn = __builtin_nanf("");
    reinterpret_cast<float4*>(out)[it] = float4{nan[0] ? qn : best[0], nan[1] ? qn : best[1], nan[2] ? qn : best[2], nan[3] ? qn : best[3]};
  }
}

}  // namespace

static inline unsigned nhwc_grid(long long items, long long per_block) {
  long long g = (items + per_block - 1) / per_block;
  if (g > 8192) g = 8192;     // grid-stride beyond that: ~32 workgroups per CU
  return (unsigned)(g < 1 ? 1 : g);
}

int launch_channel_affine_nhwc(const float* x, const float* scale, const float* shift, const float* res, const float* rscale,
                               const float* rshift, int relu, long long rows, int C, float* out, hipStream_t st) {
  if (!x || !out || rows <= 0 || C <= 0 || (C & 3)) return RMNET_E_INVALID_ARG;
  if (!res && (rscale || rshift)) return RMNET_E_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(scale) |
       reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(rscale) | reinterpret_cast<uintptr_t>(rshift)) & 15)
    return RMNET_E_INVALID_ARG;
  const long long n4 = rows * (C >> 2);
  const dim3 grid(nhwc_grid(n4, (long long)kThreads * kUnroll));
  const bool fixed = kThreads % (C >> 2) == 0;      // (every layer of both networks: C = 64 ... 1024)
#define RMNET_CAN(R, F) hipLaunchKernelGGL((channel_affine_nhwc<R, F>), grid, dim3(kThreads), 0, st, x, scale, shift, res, rscale, rshift, relu, C >> 2, n4, out)
  if (res) { if (fixed) RMNET_CAN(true, true); else RMNET_CAN(true, false); }
  else     { if (fixed) RMNET_CAN(false, true); else RMNET_CAN(false, false); }
#undef RMNET_CAN
  return check_launch();
}

int launch_upsample2x_add_nhwc(const float* x, const float* skip, long long N, int C, int h, int w, float* out, hipStream_t st) {
  if (!x || !out || N <= 0 || C <= 0 || (C & 3) || h <= 0 || w <= 0) return RMNET_E_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(skip)) & 15) return RMNET_E_INVALID_ARG;
  const long long items = N * 4 * h * w * (C >> 2);
  const dim3 grid(nhwc_grid(items, kThreads));
  if (skip)
    hipLaunchKernelGGL(upsample2x_add_nhwc<true>, grid, dim3(kThreads), 0, st, x, skip, out, h, w, C >> 2, items);
  else
    hipLaunchKernelGGL(upsample2x_add_nhwc<false>, grid, dim3(kThreads), 0, st, x, skip, out, h, w, C >> 2, items);
  return check_launch();
}

int launch_affine_relu_maxpool_nhwc(const float* x, const float* scale, const float* shift, long long N, int C, int H, int W, float* out,
                                    hipStream_t st) {
  if (!x || !out || N <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0) return RMNET_E_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15)
    return RMNET_E_INVALID_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long items = N * Ho * Wo * (C >> 2);
  hipLaunchKernelGGL(affine_relu_maxpool_nhwc, dim3(nhwc_grid(items, kThreads)), dim3(kThreads), 0, st, x, scale, shift, C >> 2, H, W, Ho, Wo,
                     items, out);
  return check_launch();
}

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

namespace rmnet {

int launch_upsample2x_add(const float* x, const float* skip, long long N, int C, int h, int w,
                          float* out, hipStream_t st) {
  if (!x || !out || N <= 0 || C <= 0 || h <= 0 || w <= 0) return RMNET_E_INVALID_ARG;
  const long long planes = N * C;
  const bool vec = (w & 1) == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(skip)) & 15) == 0;
  const long long items = (long long)2 * h * (vec ? (2 * w) >> 2 : 2 * w);
  long long chunks = (items + kThreads - 1) / kThreads;
  if (chunks > 256) chunks = 256;
  const dim3 grid((unsigned)chunks, (unsigned)(planes < 65535 ? planes : 65535));
#define RMNET_UP(V, A) hipLaunchKernelGGL((upsample2x_add<V, A>), grid, dim3(kThreads), 0, st, x, skip, out, h, w, planes)
  if (vec) { if (skip) RMNET_UP(true, true); else RMNET_UP(true, false); }
  else     { if (skip) RMNET_UP(false, true); else RMNET_UP(false, false); }
#undef RMNET_UP
  return check_launch();
}

int launch_soft_aggregate(const float* dec, const int32_t* obj_begin, int B, int K, int Hp, int Wp,
                          int pad_l, int pad_t, int H, int W, float* logit, float* prob,
                          hipStream_t st) {
  if (!dec || !obj_begin || !logit || B <= 0 || K <= 0 || H <= 0 || W <= 0) return RMNET_E_INVALID_ARG;
  if (pad_l < 0 || pad_t < 0 || pad_l + W > Wp || pad_t + H > Hp) return RMNET_E_INVALID_ARG;
  if (B > 65535) return RMNET_E_UNSUPPORTED;
  long long chunks = ((long long)H * W + kThreads - 1) / kThreads;
  if (chunks > 1024) chunks = 1024;
  hipLaunchKernelGGL(soft_aggregate, dim3((unsigned)chunks, (unsigned)B), dim3(kThreads), 0, st, dec,
                     obj_begin, K, Hp, Wp, pad_l, pad_t, H, W, logit, prob);
  return check_launch();
}

int launch_affine_relu_maxpool(const float* x, const float* scale, const float* shift, long long N,
                               int C, int H, int W, float* out, hipStream_t st) {
  if (!x || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return RMNET_E_INVALID_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long planes = N * C;
  const bool vec = W == 2 * Wo && (Wo & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  long long chunks = ((long long)Ho * (vec ? Wo >> 2 : Wo) + kThreads - 1) / kThreads;
  if (chunks > 128) chunks = 128;
  const dim3 grid((unsigned)chunks, (unsigned)(planes < 65535 ? planes : 65535));
  if (vec)
    hipLaunchKernelGGL(affine_relu_maxpool<true>, grid, dim3(kThreads), 0, st, x, scale, shift, C, planes, H, W, Ho, Wo, out);
  else
    hipLaunchKernelGGL(affine_relu_maxpool<false>, grid, dim3(kThreads), 0, st, x, scale, shift, C, planes, H, W, Ho, Wo, out);
  return check_launch();
}

}  // namespace rmnet
