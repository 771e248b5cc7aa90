/*
 * oracle/rmnet_oracle.c -- CPU restatement of RMNet's per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rmnet_amd/ may include, link, import
 * or call this file.  It exists so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg have an independent checker for the HIP kernels.
 *
 * Each function restates one reference routine (paths relative to the reference
 * checkout, hzxie/RMNet):
 *   oracle_region_map_f32      extensions/reg_att_map_generator/reg_att_map_generator.cu:15-93
 *   oracle_flow_affine_f32     extensions/flow_affine_transformation/flow_affine_transformation.cpp:63-83
 *   oracle_memory_read_f32     models/rmnet.py:147-165  (MemoryReader.forward)
 *   oracle_rect_mask_f32       models/rmnet.py:244-248, 356-358 (K/V regional masking)
 *   oracle_cell_rects_i32      models/rmnet.py:245, 356 (nearest x1/16 of a box map) restated on boxes
 *
 * Pinning (see oracle/README.md): flow_affine is checked bit-for-bit against the
 * reference C++ compiled from /root/reference (oracle/_ref); memory_read against
 * golden vectors captured from the reference's own MemoryReader; the region map's
 * box finder against the reference's own Python box finder
 * (utils/helpers.py:93-102, tests/golden/region_boxes.npz) and its loosen / clamp /
 * fallback branches by hand-computed known-answer tests (the .cu needs nvcc and a
 * CUDA device: it cannot run here).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ---------------------------------------------------------------------------------
 * G1: regional attention map.  reg_att_map_generator.cu:15-93.
 * mask [B,K,H,W] f32 -> att [B,K,H,W] f32 (0/1), bboxes [B,K,4] int32
 * (x_min, x_max, y_min, y_max).  Channel 0 is never visited: att = 0, bbox = 0
 * (outputs come from torch::zeros, .cu:104-109).
 * ------------------------------------------------------------------------------- */
void oracle_region_map_f32(const float *mask, int B, int K, int H, int W, float thr,
                           int n_pts_threshold, int loose, float *att, int32_t *bboxes) {
  const size_t npix = (size_t)H * W;
  memset(att, 0, sizeof(float) * (size_t)B * K * npix);
  memset(bboxes, 0, sizeof(int32_t) * (size_t)B * K * 4);
  for (int b = 0; b < B; ++b) {
    for (int k = 1; k < K; ++k) {
      const float *m = mask + ((size_t)b * K + k) * npix;
      float *a = att + ((size_t)b * K + k) * npix;
      int32_t *bb = bboxes + ((size_t)b * K + k) * 4;
      /* .cu:31-34: x_min/y_min start at 32767; x_max/y_max at 0 (host zeros). */
      int n = 0, x0 = 32767, x1 = 0, y0 = 32767, y1 = 0;
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
          if (m[(size_t)y * W + x] >= thr) { /* .cu:42 */
            ++n;
            if (x < x0) x0 = x;
            if (x > x1) x1 = x;
            if (y < y0) y0 = y;
            if (y > y1) y1 = y;
          }
      if (n < n_pts_threshold) { /* .cu:57-61 */
        x0 = 0; x1 = W - 1; y0 = 0; y1 = H - 1;
      } else { /* .cu:63-74; note <= and >= */
        x0 = x0 <= loose ? 0 : x0 - loose;
        x1 = x1 + loose >= W ? W - 1 : x1 + loose;
        y0 = y0 <= loose ? 0 : y0 - loose;
        y1 = y1 + loose >= H ? H - 1 : y1 + loose;
      }
      bb[0] = x0; bb[1] = x1; bb[2] = y0; bb[3] = y1;
      for (int y = 0; y < H; ++y) /* .cu:81-92, inclusive */
        for (int x = 0; x < W; ++x)
          if (x >= x0 && x <= x1 && y >= y0 && y <= y1) a[(size_t)y * W + x] = 1.0f;
    }
  }
}

/* ---------------------------------------------------------------------------------
 * Box -> 1/16-resolution cell rectangle.  models/rmnet.py:245 and :356 take the
 * full-resolution 0/1 box map (zero-padded by (lw, lh) on the query side, :307) and
 * F.interpolate(scale_factor=1/16) it with mode='nearest', which samples source pixel
 * (stride*cy, stride*cx).  A cell is therefore kept iff that pixel lies in the box.
 * rect = (cx0, cx1, cy0, cy1) inclusive; empty rect is (1,0,1,0).  Channel 0 (and any
 * channel flagged by k0_empty) is empty because its map is all zeros.
 * ------------------------------------------------------------------------------- */
static int ceil_div_pos(int a, int d) { return a <= 0 ? 0 : (a + d - 1) / d; }
static int floor_div(int a, int d) { return a >= 0 ? a / d : -((-a + d - 1) / d); }

void oracle_cell_rects_i32(const int32_t *bboxes, int B, int K, int lw, int lh, int stride,
                           int h, int w, int32_t *rects) {
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < K; ++k) {
      const int32_t *bb = bboxes + ((size_t)b * K + k) * 4;
      int32_t *r = rects + ((size_t)b * K + k) * 4;
      if (k == 0) { r[0] = 1; r[1] = 0; r[2] = 1; r[3] = 0; continue; }
      int cx0 = ceil_div_pos(bb[0] + lw, stride), cx1 = floor_div(bb[1] + lw, stride);
      int cy0 = ceil_div_pos(bb[2] + lh, stride), cy1 = floor_div(bb[3] + lh, stride);
      if (cx1 > w - 1) cx1 = w - 1;
      if (cy1 > h - 1) cy1 = h - 1;
      if (cx1 < cx0 || cy1 < cy0) { cx0 = 1; cx1 = 0; cy0 = 1; cy1 = 0; }
      r[0] = cx0; r[1] = cx1; r[2] = cy0; r[3] = cy1;
    }
}

/* ---------------------------------------------------------------------------------
 * M2/M3: multiply a [n, C, T, h, w] tensor by per-(n, t) 0/1 cell rectangles.
 * models/rmnet.py:247-248 (memory), :357-358 (query, T = 1).
 * rects [n, T, 4] = (cx0, cx1, cy0, cy1) inclusive.
 * ------------------------------------------------------------------------------- */
void oracle_rect_mask_f32(const float *x, int n, int C, int T, int h, int w,
                          const int32_t *rects, float *y) {
  const size_t hw = (size_t)h * w;
  for (int o = 0; o < n; ++o)
    for (int c = 0; c < C; ++c)
      for (int t = 0; t < T; ++t) {
        const int32_t *r = rects + ((size_t)o * T + t) * 4;
        const float *xi = x + (((size_t)o * C + c) * T + t) * hw;
        float *yo = y + (((size_t)o * C + c) * T + t) * hw;
        for (int cy = 0; cy < h; ++cy)
          for (int cx = 0; cx < w; ++cx) {
            const int in = cx >= r[0] && cx <= r[1] && cy >= r[2] && cy <= r[3];
            yo[(size_t)cy * w + cx] = in ? xi[(size_t)cy * w + cx] : xi[(size_t)cy * w + cx] * 0.0f;
          }
      }
}

/* ---------------------------------------------------------------------------------
 * M1: MemoryReader.forward.  models/rmnet.py:147-165.
 *   p   = bmm(m_key^T, q_key)            [no, THW, HW]      (:155)
 *   p   = p / sqrt(D_e)                                      (:156)
 *   p   = softmax(p, dim=1)   (over THW)                     (:157)
 *   mem = bmm(m_val, p)                  [no, D_o, HW]       (:160)
 *   out = cat([mem, q_val], dim=1)       [no, 2*D_o, HW]     (:163)
 * Tensors are fp32; contractions accumulate in double and are rounded to fp32 where the
 * reference stores an fp32 tensor (after :155, :156, :157, :160) -- torch's CPU bmm and
 * softmax use blocked/vectorised fp32 sums whose order is unspecified, so a double
 * accumulator is the neutral restatement; parity with it is a tolerance, not bit-exact.
 * p_out may be NULL.  Parallel over query cells (independent columns).
 * ------------------------------------------------------------------------------- */
void oracle_memory_read_f32(const float *m_key, const float *m_val, const float *q_key,
                            const float *q_val, int no, int De, int Do, int T, int h, int w,
                            float *mem_val, float *p_out) {
  const size_t hw = (size_t)h * w, thw = (size_t)T * hw;
  const float sqrt_de = sqrtf((float)De);
  for (int o = 0; o < no; ++o) {
    const float *mk = m_key + (size_t)o * De * thw; /* [De][THW] */
    const float *mv = m_val + (size_t)o * Do * thw; /* [Do][THW] */
    const float *qk = q_key + (size_t)o * De * hw;  /* [De][HW]  */
    const float *qv = q_val + (size_t)o * Do * hw;
    float *out = mem_val + (size_t)o * 2 * Do * hw;
    float *po = p_out ? p_out + (size_t)o * thw * hw : NULL;
#pragma omp parallel
    {
      float *col = (float *)malloc(sizeof(float) * thw);
      float *qc = (float *)malloc(sizeof(float) * De);
#pragma omp for schedule(dynamic, 8)
      for (long i = 0; i < (long)hw; ++i) {
        for (int c = 0; c < De; ++c) qc[c] = qk[(size_t)c * hw + i];
        float mx = -INFINITY;
        for (size_t j = 0; j < thw; ++j) {
          double acc = 0.0;
          for (int c = 0; c < De; ++c) acc += (double)mk[(size_t)c * thw + j] * (double)qc[c];
          float s = (float)acc;  /* :155 result is fp32 */
          s = s / sqrt_de;       /* :156 */
          col[j] = s;
          if (s > mx) mx = s;
        }
        double sum = 0.0;
        for (size_t j = 0; j < thw; ++j) { /* :157 softmax over THW */
          float e = expf(col[j] - mx);
          col[j] = e;
          sum += (double)e;
        }
        const float fsum = (float)sum;
        for (size_t j = 0; j < thw; ++j) {
          col[j] = col[j] / fsum;
          if (po) po[j * hw + i] = col[j];
        }
        for (int d = 0; d < Do; ++d) { /* :160 */
          const float *v = mv + (size_t)d * thw;
          double acc = 0.0;
          for (size_t j = 0; j < thw; ++j) acc += (double)v[j] * (double)col[j];
          out[(size_t)d * hw + i] = (float)acc;
          out[(size_t)(Do + d) * hw + i] = qv[(size_t)d * hw + i]; /* :163 */
        }
      }
      free(col);
      free(qc);
    }
  }
}

/* ---------------------------------------------------------------------------------
 * M1 on a SAMPLE of query cells: the same arithmetic as oracle_memory_read_f32 (same
 * accumulation order per element: channels ascending in a double accumulator, then the fp32
 * roundings of :155-:157, :160), for the query cells qidx[0..nq) of every object only, so that
 * BASELINE configs[4] (720p, T = 20: 72,000 memory cells x 3,600 query cells per object) can be
 * checked against the oracle in seconds.  The loops over memory cells are unit-stride here (the
 * full version walks a channel-strided column per cell), which changes no result.
 * out [no][nq][Do]: the read-out rows (the q_val half of the cat is a copy, not sampled).
 * tests/test_oracle.py checks this function bit for bit against oracle_memory_read_f32.
 * ------------------------------------------------------------------------------- */
void oracle_memory_read_sampled_f32(const float *m_key, const float *m_val, const float *q_key,
                                    int no, int De, int Do, int T, int h, int w,
                                    const int32_t *qidx, int nq, float *out) {
  const size_t hw = (size_t)h * w, thw = (size_t)T * hw;
  const float sqrt_de = sqrtf((float)De);
  for (int o = 0; o < no; ++o) {
    const float *mk = m_key + (size_t)o * De * thw;
    const float *mv = m_val + (size_t)o * Do * thw;
    const float *qk = q_key + (size_t)o * De * hw;
#pragma omp parallel
    {
      double *acc = (double *)malloc(sizeof(double) * thw);
      float *col = (float *)malloc(sizeof(float) * thw);
#pragma omp for schedule(dynamic, 1)
      for (int n = 0; n < nq; ++n) {
        const size_t i = (size_t)qidx[n];
        for (size_t j = 0; j < thw; ++j) acc[j] = 0.0;
        for (int c = 0; c < De; ++c) {
          const double q = (double)qk[(size_t)c * hw + i];
          const float *k = mk + (size_t)c * thw;
          for (size_t j = 0; j < thw; ++j) acc[j] += (double)k[j] * q;
        }
        float mx = -INFINITY;
        for (size_t j = 0; j < thw; ++j) {
          float s = (float)acc[j]; /* :155 */
          s = s / sqrt_de;         /* :156 */
          col[j] = s;
          if (s > mx) mx = s;
        }
        double sum = 0.0;
        for (size_t j = 0; j < thw; ++j) { /* :157 */
          const float e = expf(col[j] - mx);
          col[j] = e;
          sum += (double)e;
        }
        const float fsum = (float)sum;
        for (size_t j = 0; j < thw; ++j) col[j] = col[j] / fsum;
        float *dst = out + ((size_t)o * nq + n) * Do;
        for (int d = 0; d < Do; ++d) { /* :160 */
          const float *v = mv + (size_t)d * thw;
          double a = 0.0;
          for (size_t j = 0; j < thw; ++j) a += (double)v[j] * (double)col[j];
          dst[d] = (float)a;
        }
      }
      free(acc);
      free(col);
    }
  }
}

/* ---------------------------------------------------------------------------------
 * F1: updateOpticalFlow.  flow_affine_transformation.cpp:63-83.
 * flow [H,W,2] f32, m1/m2 [2,3] f32 -> out [H,W,2] f32.  All arithmetic is IEEE fp32,
 * one rounding per operation, evaluated left to right exactly as the reference's
 * expressions (built with -ffp-contract=off so no FMA is formed); size_t -> float
 * conversions are explicit.  Quirk kept: y1 uses the already-updated x1 (.cpp:72-73).
 * ------------------------------------------------------------------------------- */
void oracle_flow_affine_f32(const float *flow, const float *m1, const float *m2, int H, int W,
                            float *out) {
  const float fw = (float)(size_t)W, fh = (float)(size_t)H;
  const float fw1 = (float)(size_t)(W - 1), fh1 = (float)(size_t)(H - 1);
  for (size_t i = 0; i < (size_t)H; ++i) {
    for (size_t j = 0; j < (size_t)W; ++j) {
      const size_t idx = (i * (size_t)W + j) * 2;
      const float fj = (float)j, fi = (float)i;
      float x2 = roundf(m2[0] * fj + m2[1] * fi + m2[2]);
      float y2 = roundf(m2[3] * fj + m2[4] * fi + m2[5]);
      float x1 = fj + flow[idx];
      float y1 = fi + flow[idx + 1];
      x1 = roundf(m1[0] * x1 + m1[1] * y1 + m1[2]);
      y1 = roundf(m1[3] * x1 + m1[4] * y1 + m1[5]);
      x1 = x1 < 0 ? 0 : (x1 >= fw ? fw1 : x1);
      y1 = y1 < 0 ? 0 : (y1 >= fh ? fh1 : y1);
      x2 = x2 < 0 ? 0 : (x2 >= fw ? fw1 : x2);
      y2 = y2 < 0 ? 0 : (y2 >= fh ? fh1 : y2);
      out[idx] = x1 - x2;
      out[idx + 1] = y1 - y2;
    }
  }
}
