// bank.hip -- regional memory BANK for gfx950: append (memorise) + read (segment).
// SURVEY.md section 8 rows M1-M3 fused with P1/P2 (memory append) -- the per-frame hot path as the
// RMNet frame loop actually uses it.
//
// Why a bank.  The reference keeps memory as fp32 [B, K, C, T, h, w], re-multiplies it by 0/1 box
// maps and torch.cat's the whole thing every frame (models/rmnet.py:244-248, 416-426).  A memory
// frame's box is fixed when the frame is memorised, so the masked cells of that frame will NEVER
// contribute anything but exp(0 - m) to a later soft-max.  The bank therefore stores, per object and
// per memorised frame, ONLY the cells inside the box, already in the order and number format the
// read kernel's MFMA fragments want:
//     keys    Kh, Kl : [slot][cell n][128 channels]   fp16 hi / lo planes (cell-major = the A operand
//                      of S = K^T Q: 8 consecutive channels per lane, 16-byte loads)
//     values  Vh, Vl : [slot][512 channels][cell n']  fp16 hi / lo planes (channel-major = the A
//                      operand of O = V P), n' permuted inside every group of 32 cells so that the
//                      8 cells a lane needs are contiguous (see kperm below)
//     area           : [slot] number of cells inside the box (cells beyond it, up to the next
//                      multiple of 32, are zero padding)
//   hi = fp16(x), lo = fp16(x - hi): x = hi + lo to 2^-22 relative, so
//       a*b ~= ah*bh + ah*bl + al*bh          (three f16 MFMAs, fp32 accumulate)
//   carries fp32-class accuracy (measured: smaller error than a plain fp32 dot product)
//   at 16x/3 the rate of the fp32 MFMA.  Bytes per element are the same as fp32 (2 + 2).
//
// bk_append : one launch per memorised frame.  Reads k4/v4 (fp32, the reference's NCHW layout) once,
//             compacts to the box, splits, transposes K through LDS, permutes V, writes the slot.
// bk_main   : the read.  Workgroup = 4 waves = 64 compacted queries x one split of the tile list.
//             wave w computes S for its own 16 queries (v_mfma_f32_16x16x32_f16, K tile shared
//             through swizzled LDS, double-buffered, filled with 16-byte coalesced loads), does the
//             online soft-max in registers (a query's row lives in 4 lanes x 8 registers), splits P
//             to fp16 hi/lo in the MFMA B-fragment layout and publishes the fragments in LDS;
//             then accumulates O += V P for ITS 128 value channels x all 64 queries, with the V
//             A-fragments loaded straight from the bank into registers (16 B / lane, no LDS, no
//             transpose) one tile ahead.  ONE barrier per 32-cell tile.
// bk_combine: (memory_read.hip's mr_combine, shared) merges the splits, adds the closed-form term
//             for the masked memory cells, scatters to query cells, appends q_val * box.
#include "common.h"

namespace rmnet {
namespace {

constexpr int kDe = 128, kDo = 512;
constexpr int kQT = 64, kJT = 32;
constexpr int kThreads = 256;
constexpr int kMaxT = 512;
constexpr float kDefer = 8.0f;   // P <= e^8 stays far inside fp16 range (65504)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline void split_f16(float x, _Float16& hi, _Float16& lo) {
  const float c = fminf(fmaxf(x, -65504.0f), 65504.0f);   // saturate instead of inf/NaN
  hi = (_Float16)c;
  lo = (_Float16)fminf(fmaxf(x - (float)hi, -65504.0f), 65504.0f);
}

// Position of compact offset j (0..31) inside its 32-cell group of a V row.  The S = K^T Q MFMA
// leaves lane group g with rows {4g..4g+3} and {16+4g..16+4g+3}; storing V in that order makes
// those 8 cells one 16-byte chunk (the k index of the O = V P MFMA is then simply 8g + e).
__host__ __device__ inline int kperm(int j) {
  return 8 * ((j & 15) >> 2) + (j & 3) + 4 * (j >> 4);
}

}  // namespace

BankView bank_view(void* base, int no, int Tcap, int h, int w) {
  BankView b;
  b.no = no; b.Tcap = Tcap; b.h = h; b.w = w; b.hw = h * w;
  b.hwp = (b.hw + kJT - 1) / kJT * kJT;
  const size_t kplane = (size_t)no * Tcap * b.hwp * kDe * sizeof(_Float16);
  const size_t vplane = (size_t)no * Tcap * kDo * b.hwp * sizeof(_Float16);
  char* p = static_cast<char*>(base);
  b.kh = p; p += kplane;
  b.kl = p; p += kplane;
  b.vh = p; p += vplane;
  b.vl = p; p += vplane;
  b.area = reinterpret_cast<int32_t*>(p);
  return b;
}

size_t bank_bytes(int no, int Tcap, int h, int w) {
  const size_t hwp = ((size_t)h * w + kJT - 1) / kJT * kJT;
  return 2 * (size_t)no * Tcap * hwp * kDe * 2 + 2 * (size_t)no * Tcap * kDo * hwp * 2 +
         (((size_t)no * Tcap * 4 + 255) & ~(size_t)255);
}

namespace {

// ------------------------------------------------------------------------------------------ append
__global__ __launch_bounds__(kThreads) void bk_append(BankView b, int slot,
                                                      const float* __restrict__ k4,
                                                      const float* __restrict__ v4,
                                                      const int32_t* __restrict__ rects) {
  __shared__ float tile[kJT][kDe + 1];
  const int u = blockIdx.x, o = blockIdx.y, tid = threadIdx.x;
  Rect rc{0, b.w - 1, 0, b.h - 1};
  if (rects) {
    const int32_t* r = rects + (size_t)o * 4;
    rc = Rect{max(r[0], 0), min(r[1], b.w - 1), max(r[2], 0), min(r[3], b.h - 1)};
  }
  const int area = rc.area();
  if (u == 0 && tid == 0) b.area[(size_t)o * b.Tcap + slot] = area;
  if (u * kJT >= area) return;   // nothing of this group is ever read
  const int p = tid & 31, rg = tid >> 5;
  const int n = u * kJT + p;
  const bool valid = n < area;
  int cell = 0;
  if (valid) {
    const int rw = rc.width(), ry = n / rw;
    cell = (rc.cy0 + ry) * b.w + rc.cx0 + (n - ry * rw);
  }
  const size_t so = (size_t)o * b.Tcap + slot;
  // keys: gather [c][cell] -> LDS [p][c] -> split -> [n][c]
  const float* kb = k4 + (size_t)o * kDe * b.hw + cell;
#pragma unroll
  for (int i = 0; i < kDe / 8; ++i) {
    const int c = rg + 8 * i;
    tile[p][c] = valid ? kb[(size_t)c * b.hw] : 0.0f;
  }
  __syncthreads();
  {
    const int row = tid >> 3, c0 = (tid & 7) * 16;
    half8 h0, h1, l0, l1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 hi, lo;
      split_f16(tile[row][c0 + e], hi, lo);
      h0[e] = hi; l0[e] = lo;
      split_f16(tile[row][c0 + 8 + e], hi, lo);
      h1[e] = hi; l1[e] = lo;
    }
    const size_t off = ((so * b.hwp + (size_t)u * kJT + row) * kDe + c0) * sizeof(_Float16);
    *reinterpret_cast<half8*>(b.kh + off) = h0;
    *reinterpret_cast<half8*>(b.kh + off + 16) = h1;
    *reinterpret_cast<half8*>(b.kl + off) = l0;
    *reinterpret_cast<half8*>(b.kl + off + 16) = l1;
  }
  // values: gather [d][cell] -> split -> [d][perm(n)]
  const float* vb = v4 + (size_t)o * kDo * b.hw + cell;
  const int pp = kperm(p);
  _Float16* vh = reinterpret_cast<_Float16*>(b.vh) + (so * kDo) * b.hwp + (size_t)u * kJT + pp;
  _Float16* vl = reinterpret_cast<_Float16*>(b.vl) + (so * kDo) * b.hwp + (size_t)u * kJT + pp;
#pragma unroll 8
  for (int i = 0; i < kDo / 8; ++i) {
    const int d = rg + 8 * i;
    const float x = valid ? vb[(size_t)d * b.hw] : 0.0f;
    _Float16 hi, lo;
    split_f16(x, hi, lo);
    vh[(size_t)d * b.hwp] = hi;
    vl[(size_t)d * b.hwp] = lo;
  }
}

// ------------------------------------------------------------------------------------------ read
struct BArgs {
  BankView b;
  const float *qk, *qv;
  const int32_t* qry_rects;  // [no][4] or null
  float* ws_o;               // [no][slots][kDo][kQT]
  float* ws_ml;              // [no][slots][2][kQT]
  int T, slots;
  float inv_sqrt_de;
};

constexpr int kKbuf = kJT * kDe * 2;                       // bytes of one K plane tile (8 KB)
constexpr int kLdsBytes = 4 * kKbuf                        // K hi/lo x 2 buffers
                          + 2 * 4 * 2 * 64 * 16            // P fragments [buf][ntile][hi/lo][lane] x 16 B
                          + 2 * kQT * 4                    // alpha [buf][64]
                          + (kMaxT + 4) * 4;               // tile prefix
constexpr int kRThreads = 512;
#ifndef BK_ABLATE
#define BK_ABLATE 0   // experiments only: 1 = no V reloads, 2 = no PV MFMAs, 3 = no S/soft-max
#endif                             // 8 waves = 2 per SIMD

// Workgroup = 8 waves (2 per SIMD), 64 compacted queries x one split of the tile list.
//   waves 0-3 ("producers", static priority 2): S = K^T Q for 16 queries each, online soft-max,
//              P -> fp16 hi/lo fragments -> LDS; then their share of O += V P.
//   waves 4-7 ("consumers"): only O += V P.
//   Every wave owns 64 value channels (4 d-tiles x 4 query tiles = 64 accumulator registers) and
//   streams its V A-fragments straight from the bank, one tile ahead.  Wave i and wave i+4 share a
//   SIMD: because the producer has priority it runs PV(n) and S(n+1) first, and while it is busy
//   with the soft-max VALU work of tile n+1 the consumer's PV(n) MFMAs fill the matrix pipe.
__global__ __launch_bounds__(kRThreads, 2) void bk_main(const BArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  char* Kl_ = lds;                                 // [buf][plane][8 KB]
  char* Pl_ = lds + 4 * kKbuf;                     // [buf][ntile][plane][lane*16]
  float* Al = reinterpret_cast<float*>(Pl_ + 2 * 4 * 2 * 64 * 16);
  int* tpre = reinterpret_cast<int*>(Al + 2 * kQT);

  const BankView& b = a.b;
  const int tid = threadIdx.x, o = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const bool producer = wave < 4;
  if (producer) __builtin_amdgcn_s_setprio(2);

  // ---- plan: tile prefix over the T memorised frames, query rectangle, split decode
  if (tid < RMNET_WAVE) {
    int carry = 0;
    for (int base = 0; base < a.T; base += RMNET_WAVE) {
      const int t = base + tid;
      const int v = t < a.T ? (b.area[(size_t)o * b.Tcap + t] + kJT - 1) / kJT : 0;
      int s = v;
#pragma unroll
      for (int d = 1; d < RMNET_WAVE; d <<= 1) {
        const int up = __shfl_up(s, d);
        if (tid >= d) s += up;
      }
      if (t < a.T) tpre[t + 1] = carry + s;
      carry += __shfl(s, RMNET_WAVE - 1);
    }
    if (tid == 0) tpre[0] = 0;
  }
  __syncthreads();
  const int njt = tpre[a.T];
  Rect qr{0, b.w - 1, 0, b.h - 1};
  if (a.qry_rects) {
    const int32_t* q = a.qry_rects + (size_t)o * 4;
    qr = Rect{max(q[0], 0), min(q[1], b.w - 1), max(q[2], 0), min(q[3], b.h - 1)};
  }
  const int Mq = qr.area();
  const BankPlan pl = bank_plan(Mq, b.hw, njt, b.no, a.slots);
  const int nact = pl.nqt * pl.nsplit;
  if ((int)blockIdx.x >= nact) return;
  int L;
  {
    const int q8 = nact >> 3, r8 = nact & 7, x = blockIdx.x & 7;   // XCD-contiguous logical ids
    L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (blockIdx.x >> 3);
  }
  const int s = L / pl.nqt, qt = L - s * pl.nqt;
  const int jt0 = (int)(((long long)s * njt) / pl.nsplit);
  const int jt1 = (int)(((long long)(s + 1) * njt) / pl.nsplit);

  // Tile order: the walk may start at any tile of the split and wrap around (the online soft-max
  // does not care).  Measured on MI355X: rotating the start by the query tile to spread L2 channel
  // load is 4-20 % SLOWER (the workgroups of a split then stop sharing L2 lines in time), so the
  // rotation is off; the wrap-around walk is kept because it costs nothing.
  const int ntl = jt1 - jt0;
  auto frame_of = [&](int j) {
    int lo = 0, hi = a.T;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tpre[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
  };
  const int t_first = frame_of(jt0);
  constexpr bool kRotate = false;
  int jcur = jt0 + (kRotate ? (int)(((long long)qt * ntl) / pl.nqt) : 0);
  int t = frame_of(jcur);
  const size_t so0 = (size_t)o * b.Tcap;
  // K staging: the tile is one contiguous 8 KB block per plane; 512 threads x 16 B
  half8 kr[2];
  auto k_load = [&](int tt, int lt) {
    const size_t off = ((so0 + tt) * b.hwp + (size_t)lt * kJT) * kDe * sizeof(_Float16) + (size_t)tid * 16;
    kr[0] = *reinterpret_cast<const half8*>(b.kh + off);
    kr[1] = *reinterpret_cast<const half8*>(b.kl + off);
  };
  auto k_store = [&](int buf) {
    // row = byte / 256, chunk = (byte / 16) & 15, stored at chunk ^ (row & 15): conflict-free b128 reads
    const int row = tid >> 4, ch = tid & 15;
    char* base = Kl_ + buf * 2 * kKbuf + row * 256 + ((ch ^ (row & 15)) << 4);
    *reinterpret_cast<half8*>(base) = kr[0];
    *reinterpret_cast<half8*>(base + kKbuf) = kr[1];
  };
  // V fragments: lane (channel l15 of d-tile dt, group g) loads its 8 cells (16 B) of each plane
  half8 vh[4], vl[4];
  const size_t vrow = (size_t)(wave * 64 + l15) * b.hwp + 8 * g;
  auto v_load = [&](int dt, int tt, int lt) {
    const size_t off = (((so0 + tt) * kDo) * b.hwp + vrow + (size_t)dt * 16 * b.hwp + (size_t)lt * kJT) * sizeof(_Float16);
    vh[dt] = *reinterpret_cast<const half8*>(b.vh + off);
    vl[dt] = *reinterpret_cast<const half8*>(b.vl + off);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[dt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mref = -INFINITY, lsum = 0.0f;

  // ---- prologue: K and V loads of the first tile go out first, the scattered query loads behind
  //      them, so the HBM latencies overlap
  int lt = jcur - tpre[t];
  k_load(t, lt);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) v_load(dt, t, lt);
  // query fragments (B operand of S): lane (query l15, group g) holds channels 32ks + 8g + e
  half8 qh[4], ql[4];
  if (producer) {
    const int qn = qt * kQT + wave * 16 + l15;
    const bool qvalid = qn < Mq;
    int cell = 0;
    if (qvalid) {
      const int rw = qr.width(), ry = qn / rw;
      cell = (qr.cy0 + ry) * b.w + qr.cx0 + (qn - ry * rw);
    }
    const float* qb = a.qk + (size_t)o * kDe * b.hw + cell;
    const float keep = qvalid ? 1.0f : 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = qb[(size_t)(32 * ks + 8 * g + e) * b.hw] * keep;
        _Float16 hi, lo;
        split_f16(x, hi, lo);
        qh[ks][e] = hi; ql[ks][e] = lo;
      }
  }
  k_store(0);
  __syncthreads();

  for (int it_ = 0; it_ < ntl; ++it_) {
    const int buf = it_ & 1;
    int jn = jcur + 1, tn = t, ltn = lt + 1;
    const bool has_next = it_ + 1 < ntl;
    if (has_next) {
      if (jn == jt1) { jn = jt0; tn = t_first; }   // wrap around
      while (tpre[tn + 1] <= jn) ++tn;             // skips frames with an empty box
      ltn = jn - tpre[tn];
      k_load(tn, ltn);
    }

    if (producer) {
      const int nvalid = b.area[so0 + t] - lt * kJT;   // cells of this tile that exist (>= 1)
      // ---- S = K^T Q (hi*hi + hi*lo + lo*hi), this wave's 16 queries x 32 cells
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      const char* kb = Kl_ + buf * 2 * kKbuf;
#pragma unroll
      for (int half = 0; half < 2; ++half) {   // 8 fragment reads in flight, then their 12 MFMAs
        half8 a0h[2], a1h[2], a0l[2], a1l[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int sw = ((4 * (2 * half + k2) + g) ^ l15) << 4;
          a0h[k2] = *reinterpret_cast<const half8*>(kb + l15 * 256 + sw);
          a1h[k2] = *reinterpret_cast<const half8*>(kb + (16 + l15) * 256 + sw);
          a0l[k2] = *reinterpret_cast<const half8*>(kb + kKbuf + l15 * 256 + sw);
          a1l[k2] = *reinterpret_cast<const half8*>(kb + kKbuf + (16 + l15) * 256 + sw);
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int ks = 2 * half + k2;
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0l[k2], qh[ks], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l[k2], qh[ks], s1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h[k2], ql[ks], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h[k2], ql[ks], s1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h[k2], qh[ks], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h[k2], qh[ks], s1, 0, 0, 0);
        }
      }
      // lane holds S[cell 4g + r (+16)][query l15]; k index of the P fragment: e = r (+4)
      float sv[8];
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sv[r] = (4 * g + r < nvalid) ? s0[r] * a.inv_sqrt_de : -INFINITY;        // models/rmnet.py:156
        sv[4 + r] = (16 + 4 * g + r < nvalid) ? s1[r] * a.inv_sqrt_de : -INFINITY;
        tmax = fmaxf(tmax, fmaxf(sv[r], sv[4 + r]));
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      float alpha = 1.0f;
      if (tmax > mref + kDefer) {        // deferred running reference (first tile: mref = -inf)
        alpha = __expf(mref - tmax);
        mref = tmax;
      }
      float rs = 0.0f;
      half8 ph, plo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pv = __expf(sv[e] - mref);   // v_exp_f32: 1 ulp, one instruction
        rs += pv;
        _Float16 hi, lo;
        split_f16(pv, hi, lo);
        ph[e] = hi; plo[e] = lo;
      }
      rs += __shfl_xor(rs, 16);
      rs += __shfl_xor(rs, 32);
      lsum = lsum * alpha + rs;
      char* pb = Pl_ + ((buf * 4 + wave) * 2) * 1024 + lane * 16;
      *reinterpret_cast<half8*>(pb) = ph;
      *reinterpret_cast<half8*>(pb + 1024) = plo;
      if (g == 0) Al[buf * kQT + wave * 16 + l15] = alpha;
    }
    if (has_next) k_store(buf ^ 1);
    __syncthreads();   // the one barrier per tile: P/alpha of this tile + K of the next are visible

    // ---- O += V P for this wave's 64 value channels x 64 queries
    half8 bh[4], bl[4];
    float al[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const char* pb = Pl_ + ((buf * 4 + it) * 2) * 1024 + lane * 16;
      bh[it] = *reinterpret_cast<const half8*>(pb);
      bl[it] = *reinterpret_cast<const half8*>(pb + 1024);
      al[it] = Al[buf * kQT + it * 16 + l15];
    }
    if (__any(al[0] != 1.0f || al[1] != 1.0f || al[2] != 1.0f || al[3] != 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[dt][it] *= al[it];
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const half8 xh = vh[dt], xl = vl[dt];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, bh[it], acc[dt][it], 0, 0, 0);
        acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bl[it], acc[dt][it], 0, 0, 0);
        acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bh[it], acc[dt][it], 0, 0, 0);
      }
      if (has_next && BK_ABLATE != 1) v_load(dt, tn, ltn);   // refill for the next tile (a tile ahead)
    }
    jcur = jn;
    t = tn;
    lt = ltn;
  }

  // ---- partial (O, m, l) -> workspace slot L, layout [query][channel] (16-byte stores)
  float* wo = a.ws_o + ((size_t)o * a.slots + L) * (size_t)kDo * kQT;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
      *reinterpret_cast<f32x4*>(wo + (size_t)(it * 16 + l15) * kDo + wave * 64 + dt * 16 + 4 * g) = acc[dt][it];
  if (producer && g == 0) {
    float* wm = a.ws_ml + ((size_t)o * a.slots + L) * 2 * kQT;
    wm[wave * 16 + l15] = mref;
    wm[kQT + wave * 16 + l15] = lsum;
  }
}

}  // namespace

int launch_bank_append(void* bank, int no, int Tcap, int h, int w, int slot, const float* k4,
                       const float* v4, const int32_t* rects, hipStream_t st) {
  if (!bank || !k4 || !v4 || no <= 0 || Tcap <= 0 || h <= 0 || w <= 0 || slot < 0 || slot >= Tcap)
    return RMNET_E_INVALID_ARG;
  if (no > 65535 || Tcap > kMaxT) return RMNET_E_UNSUPPORTED;
  const BankView b = bank_view(bank, no, Tcap, h, w);
  hipLaunchKernelGGL(bk_append, dim3(b.hwp / kJT, no), dim3(kThreads), 0, st, b, slot, k4, v4, rects);
  return check_launch();
}

int launch_bank_main(const BankReadArgs& m, hipStream_t st) {
  BArgs a;
  a.b = bank_view(const_cast<void*>(m.bank), m.no, m.Tcap, m.h, m.w);
  a.qk = m.qk; a.qv = m.qv; a.qry_rects = m.qry_rects;
  a.ws_o = m.ws_o; a.ws_ml = m.ws_ml;
  a.T = m.T; a.slots = m.slots;
  a.inv_sqrt_de = 1.0f / sqrtf((float)kDe);
  hipLaunchKernelGGL(bk_main, dim3(m.slots, m.no), dim3(kRThreads), 0, st, a);
  return check_launch();
}

}  // namespace rmnet
