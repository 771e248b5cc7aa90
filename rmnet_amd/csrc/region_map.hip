// region_map.hip -- regional attention map generator for gfx950.
//
// What the reference computes (extensions/reg_att_map_generator/reg_att_map_generator.cu:15-93),
// per (batch b, channel k >= 1): n = #{mask >= thr}; (x_min, x_max, y_min, y_max) of those
// pixels, starting from (32767, 0, 32767, 0); if n < n_pts_threshold the box is the full frame,
// else it is loosened by `loose` pixels with the clamps of .cu:63-74; att = 1 inside the box.
//
// MI355X-first structure (the reference runs ONE 512-thread block per batch element with five
// global atomics per foreground pixel): the op is a streaming HBM-bound reduction followed by a
// streaming fill, so it is spread over the whole chip in two launches and uses no atomics:
//
//   region_reduce : grid (chunks, K-1, B).  A block owns a band of rows of one channel, reads it
//                   with 16-byte loads (one row per block iteration, so x needs no division),
//                   reduces {count, min/max x, min/max y} in registers -> wave shuffles -> LDS,
//                   and writes ONE 5-int partial record.  Deterministic, no init pass needed.
//   region_fill   : grid (chunks, K, B).  Every block folds the <= 256 partial records of its
//                   channel (they are L2-resident), applies the threshold/loosen/clamp rule --
//                   integer arithmetic, identical in every block -- and streams its band of the
//                   0/1 map with 16-byte stores.  Block 0 of each channel also writes the box and,
//                   optionally, the box on the 1/16 feature grid (cell rectangle) so that the
//                   regional memory read never needs the full-resolution map at all.
//
// Algorithmic HBM bytes per call: 4*B*(K-1)*H*W read + 4*B*K*H*W written (+ 16*B*K).
#include "common.h"

namespace rmnet {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 256;  // partial records per channel; folded by one wave in region_fill

struct Partial {
  int n, x0, x1, y0, y1, pad0, pad1, pad2;
};

__device__ inline void fold(int& n, int& x0, int& x1, int& y0, int& y1) {
  n = wave_sum(n);
  x0 = wave_min(x0);
  x1 = wave_max(x1);
  y0 = wave_min(y0);
  y1 = wave_max(y1);
}

template <bool VEC4>
__global__ __launch_bounds__(kThreads) void region_reduce(const float* __restrict__ mask, int K,
                                                          int H, int W, float thr,
                                                          int rows_per_chunk,
                                                          Partial* __restrict__ partials) {
  const int chunk = blockIdx.x, k = blockIdx.y + 1, b = blockIdx.z;
  const int nchunks = gridDim.x;
  const float* m = mask + ((size_t)b * K + k) * (size_t)H * W;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(H, r0 + rows_per_chunk);

  // One wave per row, the block's four waves on four rows at once, and every load of a row
  // issued before the first compare: a block pays one or two memory latencies for its band instead
  // of one per row (the row-at-a-time version of this kernel took 11 us for a 1.6 MB mask).
  const int wave_id = threadIdx.x >> 6, ln = threadIdx.x & 63;
  int n = 0, x0 = 32767, x1 = 0, y0 = 32767, y1 = 0;  // .cu:31-34 sentinels
  for (int y = r0 + wave_id; y < r1; y += kThreads / RMNET_WAVE) {
    const float* row = m + (size_t)y * W;
    int hit_lo = 32767, hit_hi = -1;
    if (VEC4) {
      const float4* row4 = reinterpret_cast<const float4*>(row);
      const int w4 = W >> 2;
      for (int base = 0; base < w4; base += 4 * RMNET_WAVE) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int x4 = base + u * RMNET_WAVE + ln;
          v[u] = x4 < w4 ? row4[x4] : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int x = (base + u * RMNET_WAVE + ln) << 2;
          const bool in = x < W;   // lanes past the row end never count, whatever the threshold
          const bool c0 = in && v[u].x >= thr, c1 = in && v[u].y >= thr,
                     c2 = in && v[u].z >= thr, c3 = in && v[u].w >= thr;  // .cu:42
          n += (int)c0 + (int)c1 + (int)c2 + (int)c3;
          if (c0 | c1 | c2 | c3) {
            const int lo = c0 ? x : (c1 ? x + 1 : (c2 ? x + 2 : x + 3));
            const int hi = c3 ? x + 3 : (c2 ? x + 2 : (c1 ? x + 1 : x));
            hit_lo = min(hit_lo, lo);
            hit_hi = max(hit_hi, hi);
          }
        }
      }
    } else {
      for (int base = 0; base < W; base += 8 * RMNET_WAVE) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int x = base + u * RMNET_WAVE + ln;
          v[u] = x < W ? row[x] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int x = base + u * RMNET_WAVE + ln;
          if (x < W && v[u] >= thr) {
            ++n;
            hit_lo = min(hit_lo, x);
            hit_hi = max(hit_hi, x);
          }
        }
      }
    }
    if (hit_hi >= 0) {
      x0 = min(x0, hit_lo);
      x1 = max(x1, hit_hi);
      y0 = min(y0, y);
      y1 = max(y1, y);
    }
  }

  fold(n, x0, x1, y0, y1);
  __shared__ int red[kThreads / RMNET_WAVE][5];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    red[wave][0] = n; red[wave][1] = x0; red[wave][2] = x1; red[wave][3] = y0; red[wave][4] = y1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < kThreads / RMNET_WAVE; ++i) {
      n += red[i][0];
      x0 = min(x0, red[i][1]);
      x1 = max(x1, red[i][2]);
      y0 = min(y0, red[i][3]);
      y1 = max(y1, red[i][4]);
    }
    Partial p{n, x0, x1, y0, y1, 0, 0, 0};
    partials[((size_t)b * K + k) * nchunks + chunk] = p;
  }
}

// ------------------------------------------------------------------------------------------------
// Flow warp fused into the reduction.  RMNet.get_att_map (models/rmnet.py:280-287)
// warps the previous soft mask along the optical flow (warp(), :252-278: base grid + flow, normalise
// to [-1, 1], two grid_samples -- the mask and a validity map of ones -- threshold the validity at
// 0.9999, multiply) only to feed the >= threshold count / min / max above; the module graph runs ~14
// kernels and three full-resolution intermediates for it.  Here a lane evaluates the warped value of
// its pixel in registers and reduces it straight away.  The arithmetic follows what PyTorch-ROCm
// executes for that Python, operation by operation:
//   v  = x + flow                       (fp32 add of the integer grid)
//   g  = 2 * v * (1 / (W - 1)) - 1      (tensor / Python-scalar is a multiply by the fp32 reciprocal
//                                        in ATen's CUDA/HIP div kernel)
//   i  = ((g + 1) / 2) * (W - 1)        grid_sampler_unnormalize, align_corners = True
//   corner weights (ix_se - ix) * (iy_se - iy) ..., accumulation in the order nw, ne, sw, se with the
//   fused multiply-adds hipcc contracts in ATen's grid_sampler_2d_kernel (RMNET_WARP_FMA = 1; the GPU
//   test compares bit for bit against torch, so a change of that build detail is caught),
//   zero padding; validity = the same sum with ones; out = value * (validity >= 0.9999).
#pragma clang fp contract(off)
#ifndef RMNET_WARP_FMA
#define RMNET_WARP_FMA 1
#endif
__device__ inline float warp_value(const float* __restrict__ m, int H, int W, int x, int y,
                                   float fx, float fy, float inv_w, float inv_h) {
  const float vx = (float)x + fx, vy = (float)y + fy;
  const float gx = 2.0f * vx * inv_w - 1.0f;
  const float gy = 2.0f * vy * inv_h - 1.0f;
  const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
  const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
  const int ix_nw = (int)floorf(ix), iy_nw = (int)floorf(iy);
  const int ix_se = ix_nw + 1, iy_se = iy_nw + 1;
  const float nw = ((float)ix_se - ix) * ((float)iy_se - iy);
  const float ne = (ix - (float)ix_nw) * ((float)iy_se - iy);
  const float sw = ((float)ix_se - ix) * (iy - (float)iy_nw);
  const float se = (ix - (float)ix_nw) * (iy - (float)iy_nw);
  float acc = 0.0f, ones = 0.0f;
  auto tap = [&](int yy, int xx, float wgt) {
    if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
      const float v = m[(size_t)yy * W + xx];
#if RMNET_WARP_FMA
      acc = __builtin_fmaf(v, wgt, acc);
#else
      acc = acc + v * wgt;
#endif
      ones = ones + wgt;
    }
  };
  tap(iy_nw, ix_nw, nw);
  tap(iy_nw, ix_se, ne);
  tap(iy_se, ix_nw, sw);
  tap(iy_se, ix_se, se);
  return acc * (ones >= 0.9999f ? 1.0f : 0.0f);
}

__global__ __launch_bounds__(kThreads) void region_reduce_warped(const float* __restrict__ mask,
                                                                 const float* __restrict__ flow,
                                                                 int K, int H, int W, float thr,
                                                                 float inv_w, float inv_h,
                                                                 int rows_per_chunk,
                                                                 Partial* __restrict__ partials,
                                                                 float* __restrict__ warped) {
  const int chunk = blockIdx.x, k = blockIdx.y + 1, b = blockIdx.z;
  const int nchunks = gridDim.x;
  const float* m = mask + ((size_t)b * K + k) * (size_t)H * W;
  const float* fxp = flow + (size_t)b * 2 * H * W;
  const float* fyp = fxp + (size_t)H * W;
  float* wout = warped ? warped + ((size_t)b * K + k) * (size_t)H * W : nullptr;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(H, r0 + rows_per_chunk);
  const int wave_id = threadIdx.x >> 6, ln = threadIdx.x & 63;
  int n = 0, x0 = 32767, x1 = 0, y0 = 32767, y1 = 0;  // .cu:31-34 sentinels
  for (int y = r0 + wave_id; y < r1; y += kThreads / RMNET_WAVE) {
    int hit_lo = 32767, hit_hi = -1;
    for (int base = 0; base < W; base += 4 * RMNET_WAVE) {
      float fx[4], fy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = base + u * RMNET_WAVE + ln;
        fx[u] = x < W ? fxp[(size_t)y * W + x] : 0.0f;
        fy[u] = x < W ? fyp[(size_t)y * W + x] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = base + u * RMNET_WAVE + ln;
        if (x < W) {
          const float v = warp_value(m, H, W, x, y, fx[u], fy[u], inv_w, inv_h);
          if (wout) wout[(size_t)y * W + x] = v;
          if (v >= thr) {
            ++n;
            hit_lo = min(hit_lo, x);
            hit_hi = max(hit_hi, x);
          }
        }
      }
    }
    if (hit_hi >= 0) {
      x0 = min(x0, hit_lo);
      x1 = max(x1, hit_hi);
      y0 = min(y0, y);
      y1 = max(y1, y);
    }
  }
  fold(n, x0, x1, y0, y1);
  __shared__ int red[kThreads / RMNET_WAVE][5];
  if (ln == 0) {
    red[wave_id][0] = n; red[wave_id][1] = x0; red[wave_id][2] = x1; red[wave_id][3] = y0; red[wave_id][4] = y1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < kThreads / RMNET_WAVE; ++i) {
      n += red[i][0];
      x0 = min(x0, red[i][1]);
      x1 = max(x1, red[i][2]);
      y0 = min(y0, red[i][3]);
      y1 = max(y1, red[i][4]);
    }
    Partial p{n, x0, x1, y0, y1, 0, 0, 0};
    partials[((size_t)b * K + k) * nchunks + chunk] = p;
  }
}
#pragma clang fp contract(fast)

// Threshold / loosen / clamp rule of .cu:55-77, integer-exact.
__device__ inline void finalize_box(int n, int& x0, int& x1, int& y0, int& y1, int H, int W,
                                    int n_pts, int loose) {
  if (n < n_pts) {
    x0 = 0; x1 = W - 1; y0 = 0; y1 = H - 1;
  } else {
    x0 = x0 <= loose ? 0 : x0 - loose;
    x1 = x1 + loose >= W ? W - 1 : x1 + loose;
    y0 = y0 <= loose ? 0 : y0 - loose;
    y1 = y1 + loose >= H ? H - 1 : y1 + loose;
  }
}

template <bool VEC4>
__global__ __launch_bounds__(kThreads) void region_fill(const Partial* __restrict__ partials,
                                                        int n_part, int K, int H, int W,
                                                        int n_pts, int loose, int rows_per_chunk,
                                                        float* __restrict__ att,
                                                        int32_t* __restrict__ bboxes,
                                                        int32_t* __restrict__ cell_rects,
                                                        int pad_l, int pad_t, int cell_stride,
                                                        int ch, int cw) {
  const int chunk = blockIdx.x, k = blockIdx.y, b = blockIdx.z;
  __shared__ int box[4];
  if (threadIdx.x < RMNET_WAVE) {  // one wave folds the partial records
    int n = 0, x0 = 32767, x1 = 0, y0 = 32767, y1 = 0;
    if (k > 0) {
      for (int i = threadIdx.x; i < n_part; i += RMNET_WAVE) {   // <= kMaxChunks / 64 records per lane
        const Partial p = partials[((size_t)b * K + k) * n_part + i];
        n += p.n;
        x0 = min(x0, p.x0); x1 = max(x1, p.x1);
        y0 = min(y0, p.y0); y1 = max(y1, p.y1);
      }
    }
    fold(n, x0, x1, y0, y1);
    if (k > 0) {
      finalize_box(n, x0, x1, y0, y1, H, W, n_pts, loose);
    } else {  // background channel: never visited by the reference -> zeros (.cu:104-109)
      x0 = 0; x1 = 0; y0 = 0; y1 = 0;
    }
    if (threadIdx.x == 0) {
      box[0] = x0; box[1] = x1; box[2] = y0; box[3] = y1;
      if (chunk == 0) {
        int32_t* bb = bboxes + ((size_t)b * K + k) * 4;
        bb[0] = x0; bb[1] = x1; bb[2] = y0; bb[3] = y1;
        if (cell_rects) {
          Rect r{1, 0, 1, 0};
          if (k > 0) r = box_to_cell_rect(x0, x1, y0, y1, pad_l, pad_t, cell_stride, ch, cw);
          int32_t* cr = cell_rects + ((size_t)b * K + k) * 4;
          cr[0] = r.cx0; cr[1] = r.cx1; cr[2] = r.cy0; cr[3] = r.cy1;
        }
      }
    }
  }
  if (att == nullptr) return;
  __syncthreads();
  const int x0 = box[0], x1 = box[1], y0 = box[2], y1 = box[3];
  const bool live = k > 0;
  float* a = att + ((size_t)b * K + k) * (size_t)H * W;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(H, r0 + rows_per_chunk);
  for (int y = r0; y < r1; ++y) {
    const bool yin = live && y >= y0 && y <= y1;
    float* row = a + (size_t)y * W;
    if (VEC4) {
      float4* row4 = reinterpret_cast<float4*>(row);
      for (int x4 = threadIdx.x; x4 < (W >> 2); x4 += kThreads) {
        const int x = x4 << 2;
        float4 v;
        v.x = (yin && x >= x0 && x <= x1) ? 1.0f : 0.0f;
        v.y = (yin && x + 1 >= x0 && x + 1 <= x1) ? 1.0f : 0.0f;
        v.z = (yin && x + 2 >= x0 && x + 2 <= x1) ? 1.0f : 0.0f;
        v.w = (yin && x + 3 >= x0 && x + 3 <= x1) ? 1.0f : 0.0f;
        row4[x4] = v;
      }
    } else {
      for (int x = threadIdx.x; x < W; x += kThreads)
        row[x] = (yin && x >= x0 && x <= x1) ? 1.0f : 0.0f;
    }
  }
}

__global__ void boxes_to_rects(const int32_t* __restrict__ bboxes, int n, int k_per_batch,
                               int pad_l, int pad_t, int stride, int ch, int cw,
                               int32_t* __restrict__ rects) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Rect r{1, 0, 1, 0};
  if (k_per_batch <= 0 || (i % k_per_batch) != 0)
    r = box_to_cell_rect(bboxes[i * 4], bboxes[i * 4 + 1], bboxes[i * 4 + 2], bboxes[i * 4 + 3],
                         pad_l, pad_t, stride, ch, cw);
  rects[i * 4] = r.cx0; rects[i * 4 + 1] = r.cx1; rects[i * 4 + 2] = r.cy0; rects[i * 4 + 3] = r.cy1;
}

// y = x * rect mask (models/rmnet.py:247-248, 357-358); one block row per (n, c, t) plane.
__global__ __launch_bounds__(kThreads) void rect_mask_kernel(const float* __restrict__ x, int C,
                                                             int T, int h, int w,
                                                             const int32_t* __restrict__ rects,
                                                             float* __restrict__ y) {
  const int plane = blockIdx.x;  // (n*C + c)*T + t
  const int t = plane % T, o = plane / (T * C);
  const int32_t* r = rects + ((size_t)o * T + t) * 4;
  const Rect rc{r[0], r[1], r[2], r[3]};
  const float* xi = x + (size_t)plane * h * w;
  float* yo = y + (size_t)plane * h * w;
  for (int i = threadIdx.x; i < h * w; i += kThreads) {
    const int cy = i / w, cx = i - cy * w;
    const float v = xi[i];
    yo[i] = rc.contains(cy, cx) ? v : v * 0.0f;  // keep the reference's x*0 semantics (NaN, -0)
  }
}

int chunks_for(int B, int K, int H) {
  // Enough blocks to cover 256 CUs a few times over, at most kMaxChunks records per channel.
  const int planes = B * (K > 1 ? K - 1 : 1);
  int chunks = (1024 + planes - 1) / planes;
  if (chunks > kMaxChunks) chunks = kMaxChunks;
  if (chunks > (H + 3) / 4) chunks = (H + 3) / 4;   // >= 4 rows per block: one per wave
  if (chunks < 1) chunks = 1;
  return chunks;
}

}  // namespace

size_t region_map_ws_bytes(int B, int K, int H, int W) {
  (void)W;
  return (size_t)B * K * chunks_for(B, K, H) * sizeof(Partial);
}

int launch_region_map(const float* mask, int B, int K, int H, int W, float thr, int n_pts,
                      int loose, float* att, int32_t* bboxes, int32_t* cell_rects, int pad_l,
                      int pad_t, int cell_stride, int ch, int cw, void* ws, size_t ws_bytes,
                      hipStream_t st) {
  return launch_region_map_warped(mask, nullptr, B, K, H, W, thr, n_pts, loose, att, bboxes, cell_rects,
                                  pad_l, pad_t, cell_stride, ch, cw, nullptr, ws, ws_bytes, st);
}

int launch_region_map_warped(const float* mask, const float* flow, int B, int K, int H, int W,
                             float thr, int n_pts, int loose, float* att, int32_t* bboxes,
                             int32_t* cell_rects, int pad_l, int pad_t, int cell_stride, int ch,
                             int cw, float* warped, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!mask || !bboxes || B <= 0 || K <= 0 || H <= 0 || W <= 0 || H > 32767 || W > 32767)
    return RMNET_E_INVALID_ARG;
  if (cell_rects && (cell_stride <= 0 || ch <= 0 || cw <= 0)) return RMNET_E_INVALID_ARG;
  if (B > 65535 || K > 65535) return RMNET_E_UNSUPPORTED;
  if (!ws || ws_bytes < region_map_ws_bytes(B, K, H, W)) return RMNET_E_WORKSPACE;
  const int chunks = chunks_for(B, K, H);
  const int rows = (H + chunks - 1) / chunks;
  Partial* partials = static_cast<Partial*>(ws);
  const bool vec_in = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(mask) & 15) == 0);
  if (warped && !flow) return RMNET_E_INVALID_ARG;
  if (K > 1 && flow) {
    dim3 g(chunks, K - 1, B);
    const float inv_w = 1.0f / (float)(W > 1 ? W - 1 : 1), inv_h = 1.0f / (float)(H > 1 ? H - 1 : 1);
    hipLaunchKernelGGL(region_reduce_warped, g, dim3(kThreads), 0, st, mask, flow, K, H, W, thr, inv_w,
                       inv_h, rows, partials, warped);
    if (int e = check_launch()) return e;
  } else if (K > 1) {
    dim3 g(chunks, K - 1, B);
    if (vec_in)
      hipLaunchKernelGGL(region_reduce<true>, g, dim3(kThreads), 0, st, mask, K, H, W, thr, rows,
                         partials);
    else
      hipLaunchKernelGGL(region_reduce<false>, g, dim3(kThreads), 0, st, mask, K, H, W, thr, rows,
                         partials);
    if (int e = check_launch()) return e;
  }
  const bool vec_out = att && (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(att) & 15) == 0);
  dim3 g2(att ? chunks : 1, K, B);
  if (vec_out)
    hipLaunchKernelGGL(region_fill<true>, g2, dim3(kThreads), 0, st, partials, chunks, K, H, W,
                       n_pts, loose, rows, att, bboxes, cell_rects, pad_l, pad_t, cell_stride, ch,
                       cw);
  else
    hipLaunchKernelGGL(region_fill<false>, g2, dim3(kThreads), 0, st, partials, chunks, K, H, W,
                       n_pts, loose, rows, att, bboxes, cell_rects, pad_l, pad_t, cell_stride, ch,
                       cw);
  return check_launch();
}

int launch_boxes_to_rects(const int32_t* bboxes, int n, int k_per_batch, int pad_l, int pad_t,
                          int stride, int ch, int cw, int32_t* rects, hipStream_t st) {
  if (!bboxes || !rects || n <= 0 || stride <= 0 || ch <= 0 || cw <= 0) return RMNET_E_INVALID_ARG;
  hipLaunchKernelGGL(boxes_to_rects, dim3((n + 255) / 256), dim3(256), 0, st, bboxes, n,
                     k_per_batch, pad_l, pad_t, stride, ch, cw, rects);
  return check_launch();
}

int launch_rect_mask(const float* x, int n, int C, int T, int h, int w, const int32_t* rects,
                     float* y, hipStream_t st) {
  if (!x || !y || !rects || n <= 0 || C <= 0 || T <= 0 || h <= 0 || w <= 0)
    return RMNET_E_INVALID_ARG;
  const long long planes = (long long)n * C * T;
  if (planes > 2147483647LL) return RMNET_E_UNSUPPORTED;
  hipLaunchKernelGGL(rect_mask_kernel, dim3((unsigned)planes), dim3(kThreads), 0, st, x, C, T, h,
                     w, rects, y);
  return check_launch();
}

}  // namespace rmnet
