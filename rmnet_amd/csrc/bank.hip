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
//     values  Vh, Vl : [slot][32-cell tile][32 channel tiles][64 lanes][8 cells] fp16 hi / lo planes:
//                      exactly the A-fragment order of O = V P (lane = channel%16 + 16*group, the 8
//                      cells of a group are those the S MFMA left in that lane group, see kperm), so a
//                      wave's fragment load is ONE contiguous 1 KB block (8 full cache lines)
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

// Reductions over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48) with the gfx950
// VALU lane swaps (v_permlane16_swap / v_permlane32_swap) instead of ds_bpermute round trips through
// the LDS: swap(x, x) leaves (even rows, even rows) / (odd rows, odd rows) resp. (low half, low half) /
// (high half, high half), so one max/add per swap finishes that level in every lane.
__device__ inline float group4_max(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ inline float group4_sum(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
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
  // values: gather [d][cell] -> split -> fragment order [tile u][d/16][lane = d%16 + 16 g][8 cells].
  // One thread builds one whole 16-byte fragment row (channel d, lane group gg): its 8 cells are the
  // compact offsets {4gg..4gg+3, 16+4gg..16+4gg+3} (kperm), fetched with 8 scalar gathers, split, and
  // written with ONE 16-byte store per plane; a wave covers 16 channels x 4 groups = 4 x 256 B runs.
  {
    const int gg = tid & 3;
    int cells[8];
    bool ok[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = 4 * gg + (e & 3) + 16 * (e >> 2);
      const int nn = u * kJT + j;
      ok[e] = nn < area;
      int cc = 0;
      if (ok[e]) {
        const int rw = rc.width(), ry = nn / rw;
        cc = (rc.cy0 + ry) * b.w + rc.cx0 + (nn - ry * rw);
      }
      cells[e] = cc;
    }
    const float* vb = v4 + (size_t)o * kDo * b.hw;
    const size_t vbase = (so * (b.hwp / kJT) + u) * (size_t)(kDo * kJT) * sizeof(_Float16);
#pragma unroll 2
    for (int i = 0; i < kDo / 64; ++i) {
      const int d = (tid >> 2) + 64 * i;
      half8 hi8, lo8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = ok[e] ? vb[(size_t)d * b.hw + cells[e]] : 0.0f;
        _Float16 hi, lo;
        split_f16(x, hi, lo);
        hi8[e] = hi; lo8[e] = lo;
      }
      const size_t off = vbase + (size_t)(((d >> 4) * 64) + (d & 15) + 16 * gg) * 16;
      *reinterpret_cast<half8*>(b.vh + off) = hi8;
      *reinterpret_cast<half8*>(b.vl + off) = lo8;
    }
  }
}

// ------------------------------------------------------------------------------------------ read
struct BArgs {
  BankView b;
  const float *qk, *qv;
  const int32_t* qry_rects;  // [no][4] or null
  float* ws_o;               // [no][slots][kDo][kQT]
  float* ws_ml;              // [no][slots][2][kQT]
  int32_t* ws_plan;          // [no][kPlanInts]
  int T, slots;
  float inv_sqrt_de;
};

constexpr int kKbuf = kJT * kDe * 2;                       // bytes of one K plane tile (8 KB)
constexpr int kLdsBytes = 6 * kKbuf                        // K hi/lo x 3 ring slots
                          + 2 * 4 * 2 * 64 * 16            // P fragments [buf][ntile][hi/lo][lane] x 16 B
                          + 2 * kQT * 4                    // alpha [buf][64]
                          + (kMaxT + 4) * 4                // tile prefix
                          + kMaxT * 4;                     // cells per frame
constexpr int kRThreads = 512;
#ifndef BK_TRACE
#define BK_TRACE 0     // experiments only: per-phase s_memtime stamps of block 0, waves 0 and 4
#endif
#ifndef BK_PDT
#define BK_PDT 3     // d-tiles of PV per producer wave (consumers take 8 - BK_PDT)
#endif
#ifndef BK_ABLATE
#define BK_ABLATE 0   // experiments only: 1 = no V reloads, 2 = no PV MFMAs, 3 = no S/soft-max
#endif                             // 8 waves = 2 per SIMD

// Workgroup = 8 waves (2 per SIMD), 64 compacted queries x one split of the tile list.
//   waves 0-3 ("producers", static priority 2): S = K^T Q for 16 queries each, online soft-max,
//              P -> fp16 hi/lo fragments -> LDS; plus a smaller share of O += V P (48 value channels).
//   waves 4-7 ("consumers"): O += V P for 80 value channels each.
//   Wave i and wave i+4 share a SIMD.  Between two barriers the producer runs PV(n) [24 MFMAs],
//   S(n+1) [24 MFMAs] and the soft-max VALU work of tile n+1, the consumer runs PV(n) [72 MFMAs]:
//   the matrix pipe sees 120 MFMAs per tile per SIMD and the producer's VALU phase hides under the
//   consumer's MFMAs.  V A-fragments stream from the bank one tile ahead (1 KB contiguous per wave
//   load); K tiles run two tiles ahead through registers into a double-buffered swizzled LDS tile.
struct Walk {          // per-workgroup constants of the tile walk (all wave-uniform)
  int jt0, ntl, qt, L, Mq;
  Rect qr;
  int t, lt;           // first tile
};

// One role of the workgroup.  PRODUCER: waves 0-3 (NDT = 3 d-tiles of PV + S + soft-max);
// consumer: waves 4-7 (NDT = 5 d-tiles of PV).  Separate instantiations keep each role's register
// set small (a shared body would keep the union of both alive: 268 spills).
template <bool PRODUCER>
__device__ inline void role_loop(const BArgs& a, const Walk& wk, char* Kl_, char* Pl_, float* Al,
                                 const int* tpre, const int* tarea, int wave, int tid, long long t_entry) {
  constexpr int NDT = PRODUCER ? BK_PDT : (8 - BK_PDT);
  const BankView& b = a.b;
  const int o = blockIdx.y;
  const int lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int jt0 = wk.jt0, ntl = wk.ntl;
  int t = wk.t, lt = wk.lt;
  // Coordinates of global tile j (>= current), clamped to the split's last tile.  Clamping makes
  // every prefetch UNCONDITIONAL (a load past the end just re-reads the last tile): with branches
  // around the loads hipcc cannot count them and falls back to s_waitcnt vmcnt(0), which drains the
  // loads issued a moment ago and puts their full latency on the critical path of every tile.
  const int jlast = jt0 + ntl - 1;
  auto advance = [&](int& tt, int& ll, int j) {
    j = min(j, jlast);
    while (tpre[tt + 1] <= j) ++tt;                // skips frames with an empty box
    ll = j - tpre[tt];
  };
  const size_t so0 = (size_t)o * b.Tcap;
  const size_t tiles_per_slot = (size_t)(b.hwp / kJT);
  // K: a tile is one contiguous 8 KB block per plane; 512 threads x 16 B.  LDS image: row = byte/256,
  // chunk = (byte/16)&15 stored at chunk ^ (row & 15) -> conflict-free ds_read_b128 of the A fragments.
  // Two register sets (A: even iterations, B: odd) so that two K tiles are in flight at any time.
  half8 krA[2], krB[2];
  auto k_load = [&](half8 (&kr)[2], int tt, int ll) {
    const size_t off = ((so0 + tt) * b.hwp + (size_t)ll * kJT) * kDe * sizeof(_Float16) + (size_t)tid * 16;
    kr[0] = *reinterpret_cast<const half8*>(b.kh + off);
    kr[1] = *reinterpret_cast<const half8*>(b.kl + off);
  };
  const int krow = tid >> 4;
  const int kdst = krow * 256 + (((tid & 15) ^ (krow & 15)) << 4);
  auto k_store = [&](const half8 (&kr)[2], int buf) {
    *reinterpret_cast<half8*>(Kl_ + buf * 2 * kKbuf + kdst) = kr[0];
    *reinterpret_cast<half8*>(Kl_ + buf * 2 * kKbuf + kKbuf + kdst) = kr[1];
  };
  // V: fragment-ordered planes; this wave's first d-tile and its lane inside a tile's 32 KB plane
  const int dt0 = PRODUCER ? BK_PDT * wave : 4 * BK_PDT + (8 - BK_PDT) * (wave - 4);
  const size_t vlane = (size_t)(dt0 * 64 + lane) * 16;
  auto v_tile = [&](int tt, int ll) { return ((so0 + tt) * tiles_per_slot + ll) * (size_t)(kDo * kJT * 2) + vlane; };
  // V fragment registers: set A holds even tiles, set B odd tiles; a set is refilled with tile n+2
  // right after PV(n) has consumed it, so every V load has two iterations to land.
  half8 vhA[NDT], vlA[NDT], vhB[NDT], vlB[NDT];
  f32x4 acc[NDT][4];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[dt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mref = -INFINITY, lsum = 0.0f;
#if BK_TRACE
  long long* trc = reinterpret_cast<long long*>(a.ws_o + ((size_t)o * a.slots + (a.slots - 1)) * (size_t)kDo * kQT) + (PRODUCER ? 0 : 1024);
  int trn = 0;
  const bool trace_on = blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
#define STAMP() do { if (trace_on && trn < 1000) trc[trn++] = (long long)__builtin_readcyclecounter(); } while (0)
  if (trace_on) trc[trn++] = t_entry;
#else
#define STAMP() do {} while (0)
#endif
  STAMP();

  // ---- prologue: K of the first two tiles and V of the first go out first, the scattered query
  //      loads behind them: all their latencies overlap
  int tn = t, ltn = lt;                            // tile n+1
  k_load(krA, t, lt);
  advance(tn, ltn, jt0 + 1);
  k_load(krB, tn, ltn);
  {
    const size_t off = v_tile(t, lt);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      vhA[dt] = *reinterpret_cast<const half8*>(b.vh + off + dt * 1024);
      vlA[dt] = *reinterpret_cast<const half8*>(b.vl + off + dt * 1024);
    }
    const size_t off1 = v_tile(tn, ltn);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      vhB[dt] = *reinterpret_cast<const half8*>(b.vh + off1 + dt * 1024);
      vlB[dt] = *reinterpret_cast<const half8*>(b.vl + off1 + dt * 1024);
    }
  }
  // query fragments (B operand of S): lane (query l15, group g) holds channels 32ks + 8g + e
  half8 qh[PRODUCER ? 4 : 1], ql[PRODUCER ? 4 : 1];
  if (PRODUCER) {
    const int qn = wk.qt * kQT + wave * 16 + l15;
    const bool qvalid = qn < wk.Mq;
    int cell = 0;
    if (qvalid) {
      const int rw = wk.qr.width(), ry = qn / rw;
      cell = (wk.qr.cy0 + ry) * b.w + wk.qr.cx0 + (qn - ry * rw);
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
        qh[PRODUCER ? ks : 0][e] = hi; ql[PRODUCER ? ks : 0][e] = lo;
      }
  }
  // ---- software pipeline (one barrier per tile):
  //   iteration i :  all waves      PV(i)            reads P[i&1] (published by the barrier of i-1)
  //                  producers      S(i+1), soft-max writes P[(i+1)&1]  -- one tile AHEAD of the PV
  //                  all waves      K tile i+2 (registers, requested during i-2) -> LDS ring slot
  //                                 (i+2)%3, then request K tile i+4 into the same registers
  //   so the consumers never wait for the producers' soft-max, the producers' VALU phase runs under
  //   the consumers' MFMAs, and no global-memory latency sits between a barrier and the MFMAs.
  k_store(krA, 0);
  k_store(krB, 1);
  int t2 = tn, lt2 = ltn;                          // tile n+2 (in flight in set A)
  advance(t2, lt2, jt0 + 2);
  k_load(krA, t2, lt2);
  int t3 = t2, lt3 = lt2;                          // tile n+3 (in flight in set B)
  advance(t3, lt3, jt0 + 3);
  k_load(krB, t3, lt3);

  // S = K^T Q, online soft-max and P fragments of one tile (producers only)
  auto s_phase = [&](int tt, int ll, int kslot, int pbuf) {
    const int nvalid = tarea[tt] - ll * kJT;   // cells of this tile that exist (>= 1)
    // Four independent accumulator chains: two MFMAs on one accumulator are >= 4 issues apart.
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    const char* kb = Kl_ + kslot * 2 * kKbuf;
    {   // all 16 fragment reads in flight, then the 24 MFMAs (producers have the registers for it)
      half8 a0h[4], a1h[4], a0l[4], a1l[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int sw = ((4 * ks + g) ^ l15) << 4;
        a0h[ks] = *reinterpret_cast<const half8*>(kb + l15 * 256 + sw);
        a1h[ks] = *reinterpret_cast<const half8*>(kb + (16 + l15) * 256 + sw);
        a0l[ks] = *reinterpret_cast<const half8*>(kb + kKbuf + l15 * 256 + sw);
        a1l[ks] = *reinterpret_cast<const half8*>(kb + kKbuf + (16 + l15) * 256 + sw);
      }
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int ks = PRODUCER ? k4 : 0;
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0l[k4], qh[ks], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l[k4], qh[ks], c1, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h[k4], qh[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h[k4], qh[ks], s1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h[k4], ql[ks], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h[k4], ql[ks], c1, 0, 0, 0);
      }
    }
    s0 += c0;
    s1 += c1;
    // lane holds S[cell 4g + r (+16)][query l15]; k index of the P fragment: e = r (+4)
    float sv[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sv[r] = (4 * g + r < nvalid) ? s0[r] * a.inv_sqrt_de : -INFINITY;        // models/rmnet.py:156
      sv[4 + r] = (16 + 4 * g + r < nvalid) ? s1[r] * a.inv_sqrt_de : -INFINITY;
      tmax = fmaxf(tmax, fmaxf(sv[r], sv[4 + r]));
    }
    tmax = group4_max(tmax);
    float alpha = 1.0f;
    if (tmax > mref + kDefer) {        // deferred running reference (first tile: mref = -inf)
      alpha = __expf(mref - tmax);
      mref = tmax;
    }
    float rs = 0.0f;
    half8 ph, plo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pv = __expf(sv[e] - mref);   // in [0, e^8]: no saturation needed for the split
      rs += pv;
      const _Float16 hi = (_Float16)pv;
      ph[e] = hi;
      plo[e] = (_Float16)(pv - (float)hi);
    }
    rs = group4_sum(rs);
    lsum = lsum * alpha + rs;
    char* pb = Pl_ + ((pbuf * 4 + wave) * 2) * 1024 + lane * 16;
    *reinterpret_cast<half8*>(pb) = ph;
    *reinterpret_cast<half8*>(pb + 1024) = plo;
    if (g == 0) Al[pbuf * kQT + wave * 16 + l15] = alpha;
  };

  __syncthreads();                                   // K tiles 0 and 1 visible
  if (PRODUCER && BK_ABLATE != 3) s_phase(t, lt, 0, 0);
  __syncthreads();                                   // P(0) visible
  STAMP();

  int ks1 = 1, ks2 = 2;                              // ring slots of tiles n+1 and n+2
  auto iteration = [&](const int it_, half8 (&kr)[2], half8 (&vh)[NDT], half8 (&vl)[NDT]) {
    const int buf = it_ & 1;
    STAMP();   // loop top
    const bool has_next = it_ + 1 < ntl;
    // ---- O += V P for this wave's NDT d-tiles x 64 queries (tile n)
    {
      const size_t noff = v_tile(t2, lt2);   // refill target: tile n+2 (clamped past the end)
      const char* nvh = b.vh + noff;
      const char* nvl = b.vl + noff;
      const char* pfr = Pl_ + (buf * 4) * 2048;
      float al[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) al[it] = Al[buf * kQT + it * 16 + l15];
      // P fragments are consumed one query tile at a time (16 registers live instead of 32); the
      // NDT accumulators touched between two uses of the same accumulator keep the MFMAs independent.
      half8 bh = *reinterpret_cast<const half8*>(pfr + lane * 16);
      half8 bl = *reinterpret_cast<const half8*>(pfr + 1024 + lane * 16);
      if (__any(al[0] != 1.0f || al[1] != 1.0f || al[2] != 1.0f || al[3] != 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int it = 0; it < 4; ++it) acc[dt][it] *= al[it];
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const half8 ch = bh, cl = bl;
        if (it < 3) {   // next query tile's fragments: their LDS latency hides under this tile's MFMAs
          bh = *reinterpret_cast<const half8*>(pfr + (it + 1) * 2048 + lane * 16);
          bl = *reinterpret_cast<const half8*>(pfr + (it + 1) * 2048 + 1024 + lane * 16);
        }
#if BK_ABLATE != 2
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
          acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[dt], ch, acc[dt][it], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
          acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dt], cl, acc[dt][it], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
          acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dt], ch, acc[dt][it], 0, 0, 0);
#endif
      }
      if (BK_ABLATE != 1) {   // refill this set with tile n+2 (two tiles ahead), unconditionally
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          vh[dt] = *reinterpret_cast<const half8*>(nvh + dt * 1024);
          vl[dt] = *reinterpret_cast<const half8*>(nvl + dt * 1024);
        }
      }
    }
    STAMP();   // PV done
    // ---- producers: S and soft-max of tile n+1 (its K tile became visible with the last barrier).
    //      (Doing this BEFORE the producers' own PV share was measured 4 % slower.)
    if (PRODUCER && has_next && BK_ABLATE != 3) s_phase(tn, ltn, ks1, buf ^ 1);
    STAMP();   // S/soft-max done
    // ---- K ring: tile n+2 (requested two iterations ago) -> slot ks2; request tile n+4 into the
    //      same registers (the other set holds tile n+3, still in flight)
    int t4 = t3, lt4 = lt3;
    k_store(kr, ks2);                                // (a clamped duplicate past the end: harmless)
    advance(t4, lt4, jt0 + it_ + 4);
    k_load(kr, t4, lt4);
    STAMP();   // before barrier
    __syncthreads();   // the one barrier per tile
    STAMP();   // after barrier
    t = tn; lt = ltn;
    tn = t2; ltn = lt2;
    t2 = t3; lt2 = lt3;
    t3 = t4; lt3 = lt4;
    ks1 = ks2;
    ks2 = ks2 == 2 ? 0 : ks2 + 1;
  };
  for (int it_ = 0; it_ < ntl; it_ += 2) {
    iteration(it_, krA, vhA, vlA);
    if (it_ + 1 < ntl) iteration(it_ + 1, krB, vhB, vlB);
  }
  STAMP();

  // ---- partial (O, m, l) -> workspace slot L, layout [query][channel] (16-byte stores)
  float* wo = a.ws_o + ((size_t)o * a.slots + wk.L) * (size_t)kDo * kQT;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
      *reinterpret_cast<f32x4*>(wo + (size_t)(it * 16 + l15) * kDo + (dt0 + dt) * 16 + 4 * g) = acc[dt][it];
  if (PRODUCER && g == 0) {
    float* wm = a.ws_ml + ((size_t)o * a.slots + wk.L) * 2 * kQT;
    wm[wave * 16 + l15] = mref;
    wm[kQT + wave * 16 + l15] = lsum;
  }
#if BK_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  STAMP();   // epilogue stores drained
}

__global__ __launch_bounds__(kRThreads, 2) void bk_main(const BArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  char* Kl_ = lds;                                 // [ring slot][plane][8 KB]
  char* Pl_ = lds + 6 * kKbuf;                     // [buf][ntile][plane][lane*16]
  float* Al = reinterpret_cast<float*>(Pl_ + 2 * 4 * 2 * 64 * 16);
  int* tpre = reinterpret_cast<int*>(Al + 2 * kQT);
  int* tarea = tpre + kMaxT + 4;

  const long long t_entry = (long long)__builtin_readcyclecounter();
  const BankView& b = a.b;
  const int tid = threadIdx.x, o = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
#ifndef BK_PRIO
#define BK_PRIO 2
#endif
  if (producer && BK_PRIO > 0) __builtin_amdgcn_s_setprio(BK_PRIO);

  // ---- plan: tile prefix over the T memorised frames, query rectangle, split decode.  The query
  //      rectangle load is issued first so that its latency overlaps the area loads.
  int q0 = 0, q1 = b.w - 1, q2 = 0, q3 = b.h - 1;
  if (a.qry_rects) {
    const int32_t* q = a.qry_rects + (size_t)o * 4;
    q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
  }
  if (tid < RMNET_WAVE) {
    int carry = 0;
    for (int base = 0; base < a.T; base += RMNET_WAVE) {
      const int t = base + tid;
      const int ar = t < a.T ? b.area[(size_t)o * b.Tcap + t] : 0;
      int s = (ar + kJT - 1) / kJT;
#pragma unroll
      for (int d = 1; d < RMNET_WAVE; d <<= 1) {
        const int up = __shfl_up(s, d);
        if (tid >= d) s += up;
      }
      if (t < a.T) { tpre[t + 1] = carry + s; tarea[t] = ar; }
      carry += __shfl(s, RMNET_WAVE - 1);
    }
    if (tid == 0) tpre[0] = 0;
  }
  __syncthreads();
  const int njt = tpre[a.T];
  Walk wk;
  wk.qr = Rect{max(q0, 0), min(q1, b.w - 1), max(q2, 0), min(q3, b.h - 1)};
  wk.Mq = wk.qr.area();
  const BankPlan pl = bank_plan(wk.Mq, b.hw, njt, b.no, a.slots);
  const int nact = pl.nqt * pl.nsplit;
  if (blockIdx.x == 0 && tid == 0) {   // plan record for the combine kernel
    int m = 0;
    for (int t = 0; t < a.T; ++t) m += tarea[t];
    int32_t* pr = a.ws_plan + (size_t)o * kPlanInts;
    pr[0] = wk.Mq; pr[1] = pl.nqt; pr[2] = pl.nsplit; pr[3] = m;
    pr[4] = wk.qr.cx0; pr[5] = wk.qr.cx1; pr[6] = wk.qr.cy0; pr[7] = wk.qr.cy1;
  }
  if ((int)blockIdx.x >= nact) return;
  {
    const int q8 = nact >> 3, r8 = nact & 7, x = blockIdx.x & 7;   // XCD-contiguous logical ids
    wk.L = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (blockIdx.x >> 3);
  }
  const int s = wk.L / pl.nqt;
  wk.qt = wk.L - s * pl.nqt;
  wk.jt0 = (int)(((long long)s * njt) / pl.nsplit);
  wk.ntl = (int)(((long long)(s + 1) * njt) / pl.nsplit) - wk.jt0;
  {
    int lo = 0, hi = a.T;   // frame of the first tile
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tpre[mid] <= wk.jt0) lo = mid; else hi = mid;
    }
    wk.t = lo;
    wk.lt = wk.jt0 - tpre[lo];
  }
  if (producer)
    role_loop<true>(a, wk, Kl_, Pl_, Al, tpre, tarea, wave, tid, t_entry);
  else
    role_loop<false>(a, wk, Kl_, Pl_, Al, tpre, tarea, wave, tid, t_entry);
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
  a.ws_o = m.ws_o; a.ws_ml = m.ws_ml; a.ws_plan = m.ws_plan;
  a.T = m.T; a.slots = m.slots;
  a.inv_sqrt_de = 1.0f / sqrtf((float)kDe);
  hipLaunchKernelGGL(bk_main, dim3(m.slots, m.no), dim3(kRThreads), 0, st, a);
  return check_launch();
}

}  // namespace rmnet
