// Shared device helpers for librmnet_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rmnet_hip.h"

#define RMNET_WAVE 64

namespace rmnet {

// Cell rectangle (cx0, cx1, cy0, cy1), inclusive; empty = (1, 0, 1, 0).
struct Rect {
  int cx0, cx1, cy0, cy1;
  __host__ __device__ int width() const { return cx1 >= cx0 ? cx1 - cx0 + 1 : 0; }
  __host__ __device__ int height() const { return cy1 >= cy0 ? cy1 - cy0 + 1 : 0; }
  __host__ __device__ int area() const { return width() * height(); }
  __host__ __device__ bool contains(int cy, int cx) const {
    return cx >= cx0 && cx <= cx1 && cy >= cy0 && cy <= cy1;
  }
};

// Pixel box (x_min, x_max, y_min, y_max) of a full-resolution 0/1 map that is zero-padded by
// (pad_l, pad_t) and then nearest-downsampled by `stride` (sample = pixel (stride*cy, stride*cx)
// of the padded map): the surviving cells.  models/rmnet.py:245, 307, 356.
__host__ __device__ inline Rect box_to_cell_rect(int x0, int x1, int y0, int y1, int pad_l,
                                                 int pad_t, int stride, int ch, int cw) {
  Rect r;
  const int ax0 = x0 + pad_l, ax1 = x1 + pad_l, ay0 = y0 + pad_t, ay1 = y1 + pad_t;
  r.cx0 = ax0 <= 0 ? 0 : (ax0 + stride - 1) / stride;
  r.cy0 = ay0 <= 0 ? 0 : (ay0 + stride - 1) / stride;
  r.cx1 = ax1 >= 0 ? ax1 / stride : -1;
  r.cy1 = ay1 >= 0 ? ay1 / stride : -1;
  if (r.cx1 > cw - 1) r.cx1 = cw - 1;
  if (r.cy1 > ch - 1) r.cy1 = ch - 1;
  if (r.cx1 < r.cx0 || r.cy1 < r.cy0) { r.cx0 = 1; r.cx1 = 0; r.cy0 = 1; r.cy1 = 0; }
  return r;
}

__device__ inline int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ inline int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// Launchers (one per .hip file); each returns RMNET_OK or a negative code.
int launch_region_map(const float* mask, int B, int K, int H, int W, float thr, int n_pts,
                      int loose, float* att, int32_t* bboxes, int32_t* cell_rects, int pad_l,
                      int pad_t, int cell_stride, int ch, int cw, void* ws, size_t ws_bytes,
                      hipStream_t st);
size_t region_map_ws_bytes(int B, int K, int H, int W);
int launch_boxes_to_rects(const int32_t* bboxes, int n, int k_per_batch, int pad_l, int pad_t,
                          int stride, int ch, int cw, int32_t* rects, hipStream_t st);
int launch_rect_mask(const float* x, int n, int C, int T, int h, int w, const int32_t* rects,
                     float* y, hipStream_t st);
int launch_flow_affine(const float* flow, const float* m1, const float* m2, int H, int W,
                       float* out, hipStream_t st);

struct MemReadArgs {
  const float *mk, *mv, *qk, *qv;
  float* out;
  float* p_out;
  const int32_t* mem_rects;
  const int32_t* qry_rects;
  int no, De, Do, T, h, w;
  long long mk_cs, mk_os, mv_cs, mv_os;
  int flags;
  void* ws;
  size_t ws_bytes;
  hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_end = nullptr;  // optional profiling hooks
};
size_t memory_read_ws_bytes(int no, int De, int Do, int T, int h, int w, int flags);
int launch_memory_read(const MemReadArgs& a, hipStream_t st);

inline int check_launch() { return hipGetLastError() == hipSuccess ? RMNET_OK : RMNET_E_LAUNCH; }

}  // namespace rmnet
