// memory_read.hip -- fused regional memory read for gfx950.
//
// Reference semantics (models/rmnet.py:147-165, with the box masking of :244-248 / :356-358):
//     S[j,i]   = sum_c m_key[c,j] * q_key[c,i] / sqrt(De)        j in T*h*w memory cells, i in h*w query cells
//     p[:,i]   = softmax_j S[j,i]
//     mem[d,i] = sum_j m_val[d,j] * p[j,i]
//     out      = cat(mem, q_val)
// where K,V (memory side) and q_key,q_val (query side) have been multiplied by 0/1 box maps.
// The reference materialises p ([no, THW, HW] fp32) and streams it through HBM about six times.
//
// This file never writes p (unless asked for) and never touches a masked cell:
//   * a masked MEMORY cell has key = 0 and value = 0, so S = 0 and it adds exp(0 - m) to the
//     soft-max denominator and nothing to the numerator.  All N_out such cells are folded into one
//     closed-form term N_out * exp(-m) in the combine step;
//   * a masked QUERY cell has q_key = 0, so every S is 0, the soft-max is uniform and its read-out is
//     the mean of m_val over all T*h*w cells -- the same vector for every masked query.  It is
//     produced by one zero-key "mean slot" that rides along in the last query tile.
//   So the kernel iterates over the COMPACTED unmasked memory cells x COMPACTED unmasked query
//   cells only; rectangles are read on the device (no host sync, graph-capturable).
//
// mr_main (hand-tiled for wave64 / MFMA / LDS):
//   workgroup = 4 waves = 64 queries x one split of the compacted memory axis.
//   wave w : owns queries [16w, 16w+16) for the affinity GEMM  S = K^T Q  (v_mfma_f32_16x16x4_f32,
//            M = memory cell, N = query, K = channel) -- the soft-max row (over memory cells) of a
//            query lives in 4 lanes x 8 registers, so max/sum are register ops + two shuffles;
//            owns value channels [128w, 128w+128) for O += V P  (M = channel, N = all 64 queries,
//            K = memory cell) -- 8x4 accumulator tiles = 128 registers.
//   LDS    : K tile [128][32], V tile [512][32] (row strides 48 / 34 floats = conflict-free for
//            the MFMA operand reads), P tile [32][64] exchanged between the waves, alpha[64].
//   HBM    : each K/V row segment is read with coalesced 128-byte wave transactions directly in
//            the reference's channel-major layout (positions contiguous) -- no transpose pass;
//            the next tile's loads are issued before the current tile's MFMAs (register staging).
//   soft-max: online, with a deferred running reference (the accumulators are only rescaled when
//            a tile's max exceeds the reference by > kDefer; fp32 has the range for it), so the
//            128-register rescale almost never runs.
//   occupancy: 1620 query cells are only 26 tiles, so the memory axis is split (flash-decoding
//            style) until the launch has about kTargetSlots workgroups; partial (O, m, l) go to a
//            workspace and mr_combine merges them, adds the closed-form term, scatters the mean
//            vector to masked cells and appends q_val (x box mask) -- the cat of :163.
//   XCD    : workgroups that share a memory split (same K/V bytes) are mapped to the same XCD so
//            the split is fetched from HBM once per L2.
//
// Algorithmic bytes per object-frame: 4*[(De+Do)*T*h*w + (De+Do)*h*w + 2*Do*h*w].
#include "common.h"

namespace rmnet {
namespace {

constexpr int kDe = 128, kDo = 512;
constexpr int kQT = 64;          // queries per workgroup
constexpr int kJT = 32;          // memory cells per tile
constexpr int kKS = 48;          // K tile row stride (floats): 32 + 16 -> lane groups on disjoint banks
constexpr int kVS = 34;          // V tile row stride: 16 rows x {0,1} -> 32 distinct banks
constexpr int kPS = 80;          // P tile row stride
constexpr int kMaxT = 2048;       // memorised frames per read (LDS prefix array); models/rmnet.py:416-426 has no bound
constexpr int kMaxSplits = kSplitMax;
constexpr int kTargetSlots = kSplitTargetSlots;
constexpr float kDefer = 30.0f;
constexpr int kThreads = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct KArgs {
  const float *mk, *mv, *qk, *qv;
  float* out;
  const int32_t* mem_rects;  // [no][T][4] or null
  const int32_t* qry_rects;  // [no][4] or null
  float* ws_o;               // [no][slots] partial blocks in fragment order (common.h)
  float* ws_ml;              // [no][slots][2][kQT]
  int32_t* ws_plan;          // [no][kPlanInts], written by block 0 of the read kernel
  int no, T, h, w, hw;
  long long mk_cs, mk_os, mv_cs, mv_os;
  int slots;                 // workgroup slots per object (grid.x of mr_main)
  float sqrt_de;
  const int32_t* gate;       // non-null: the overflow word of a transient bank (drop-in entry): mr_main and
                             // mr_combine run only when it is set (bk_main has then returned at once)
};

struct Plan {
  int Mq;      // unmasked query cells
  int nqt;     // query tiles (incl. the mean slot when Mq < hw)
  int M;       // unmasked memory cells
  int njt;     // memory tiles
  int nsplit;  // splits of the memory axis (0 when M == 0)
  Rect qr;
};

// Prefix sums of per-frame rectangle areas into LDS prefix[0..T]; returns the launch plan.
// Must be called by all threads of the block (contains barriers).
template <bool REGIONAL>
__device__ inline Plan make_plan(const KArgs& a, int o, int* prefix) {
  Plan p;
  const int tid = threadIdx.x;
  if (REGIONAL && a.mem_rects) {
    if (tid < RMNET_WAVE) {
      int carry = 0;
      for (int base = 0; base < a.T; base += RMNET_WAVE) {
        const int t = base + tid;
        int v = 0;
        if (t < a.T) {
          const int32_t* r = a.mem_rects + ((size_t)o * a.T + t) * 4;
          Rect rc{r[0], r[1], r[2], r[3]};
          rc.cx0 = max(rc.cx0, 0); rc.cy0 = max(rc.cy0, 0);
          rc.cx1 = min(rc.cx1, a.w - 1); rc.cy1 = min(rc.cy1, a.h - 1);
          v = rc.area();
        }
        int s = v;
#pragma unroll
        for (int d = 1; d < RMNET_WAVE; d <<= 1) {
          const int u = __shfl_up(s, d);
          if (tid >= d) s += u;
        }
        if (t < a.T) prefix[t + 1] = carry + s;
        carry += __shfl(s, RMNET_WAVE - 1);
      }
      if (tid == 0) prefix[0] = 0;
    }
    __syncthreads();
    p.M = prefix[a.T];
  } else {
    p.M = a.T * a.hw;
  }
  if (REGIONAL) {
    const int32_t* q = a.qry_rects + (size_t)o * 4;
    p.qr = Rect{max(q[0], 0), min(q[1], a.w - 1), max(q[2], 0), min(q[3], a.h - 1)};
    p.Mq = p.qr.area();
  } else {
    p.Mq = a.hw;
    p.qr = Rect{0, a.w - 1, 0, a.h - 1};
  }
  p.njt = (p.M + kJT - 1) / kJT;
  const BankPlan bp = bank_plan(p.Mq, a.hw, p.njt, a.no, a.slots);
  p.nqt = bp.nqt;
  p.nsplit = bp.nsplit;
  return p;
}

// Compacted memory index n -> element offset t*hw + cy*w + cx (REGIONAL) or n (dense).
template <bool REGIONAL>
__device__ inline int mem_index(const KArgs& a, int o, const int* prefix, int n) {
  if (!REGIONAL) return n;
  int lo = 0, hi = a.T;  // find t with prefix[t] <= n < prefix[t+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= n) lo = mid; else hi = mid;
  }
  const int32_t* r = a.mem_rects + ((size_t)o * a.T + lo) * 4;
  const int cx0 = max(r[0], 0), cx1 = min(r[1], a.w - 1), cy0 = max(r[2], 0);
  const int rw = cx1 - cx0 + 1;
  const int rem = n - prefix[lo];
  const int ry = rem / rw;
  return lo * a.hw + (cy0 + ry) * a.w + cx0 + (rem - ry * rw);
}

__device__ inline int query_cell(const Plan& p, int w, int n) {
  const int rw = p.qr.width();
  const int ry = n / rw;
  return (p.qr.cy0 + ry) * w + p.qr.cx0 + (n - ry * rw);
}

// Block b of a launch lands on XCD b % 8; give each XCD a contiguous range of logical ids so that
// workgroups sharing a memory split share an L2 (bijective for any count).
__device__ inline int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

constexpr int kLdsFloats = kDe * kKS + kDo * kVS + kJT * kPS + kQT + 4 + (kMaxT + 4);
static_assert(kLdsFloats * 4 + 2048 <= kLdsBytesPerCU, "mr_main's LDS (tiles + the prefix array of kMaxT frames) must fit one CU");

template <bool REGIONAL>
__global__ __launch_bounds__(kThreads, 1) void mr_main(const KArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  float* Kt = lds;
  float* Vt = Kt + kDe * kKS;
  float* Pt = Vt + kDo * kVS;
  float* Al = Pt + kJT * kPS;
  int* flags = reinterpret_cast<int*>(Al + kQT);
  int* prefix = flags + 4;

  const int tid = threadIdx.x, o = blockIdx.y;
  const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  if (a.gate && __builtin_amdgcn_readfirstlane(*a.gate) == 0) return;   // the split-fp16 bank path did the read
  if (tid < 2) flags[tid] = 0;
  const Plan pl = make_plan<REGIONAL>(a, o, prefix);
  const int nact = pl.nqt * pl.nsplit;
  if (blockIdx.x == 0 && tid == 0) {   // plan record for the combine kernel
    int32_t* pr = a.ws_plan + (size_t)o * kPlanInts;
    pr[0] = pl.Mq; pr[1] = pl.nqt; pr[2] = pl.nsplit; pr[3] = pl.M;
    pr[4] = pl.qr.cx0; pr[5] = pl.qr.cx1; pr[6] = pl.qr.cy0; pr[7] = pl.qr.cy1;
    pr[8] = o * a.slots; pr[9] = 0; pr[10] = 0; pr[11] = 0;   // (mode 0, common.h)
  }
  if ((int)blockIdx.x >= nact) return;
  const int L = xcd_remap(blockIdx.x, nact);
  const int s = L / pl.nqt, qt = L - s * pl.nqt;
  const int jt0 = (int)(((long long)s * pl.njt) / pl.nsplit);
  const int jt1 = (int)(((long long)(s + 1) * pl.njt) / pl.nsplit);

  // ---- query fragment (B operand of S = K^T Q): lane (l15, g) holds Q[c = 4ks + g][query l15]
  float qreg[kDe / 4];
  {
    const int qn = qt * kQT + wave * 16 + l15;
    const bool qvalid = qn < pl.Mq;
    const int cell = qvalid ? (REGIONAL ? query_cell(pl, a.w, qn) : qn) : 0;
    const float* qb = a.qk + (size_t)o * kDe * a.hw + cell;
#pragma unroll
    for (int ks = 0; ks < kDe / 4; ++ks) qreg[ks] = qvalid ? qb[(size_t)(4 * ks + g) * a.hw] : 0.0f;
  }

  // ---- staging registers: thread (jcol, rg) carries rows rg + 8k of column jcol
  const int jcol = tid & 31, rg = tid >> 5;
  float kst[kDe / 8], vst[kDo / 8];
  const float* mkb = a.mk + (size_t)o * a.mk_os + (size_t)rg * a.mk_cs;
  const float* mvb = a.mv + (size_t)o * a.mv_os + (size_t)rg * a.mv_cs;
  auto stage_load = [&](int jt) {
    const int n = jt * kJT + jcol;
    if (n < pl.M) {
      const int jg = mem_index<REGIONAL>(a, o, prefix, n);
#pragma unroll
      for (int k = 0; k < kDe / 8; ++k) kst[k] = mkb[(size_t)(8 * k) * a.mk_cs + jg];
#pragma unroll
      for (int k = 0; k < kDo / 8; ++k) vst[k] = mvb[(size_t)(8 * k) * a.mv_cs + jg];
    } else {
#pragma unroll
      for (int k = 0; k < kDe / 8; ++k) kst[k] = 0.0f;
#pragma unroll
      for (int k = 0; k < kDo / 8; ++k) vst[k] = 0.0f;
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int k = 0; k < kDe / 8; ++k) Kt[(rg + 8 * k) * kKS + jcol] = kst[k];
#pragma unroll
    for (int k = 0; k < kDo / 8; ++k) Vt[(rg + 8 * k) * kVS + jcol] = vst[k];
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[dt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mref = -INFINITY, lsum = 0.0f;

  stage_load(jt0);
  for (int jt = jt0; jt < jt1; ++jt) {
    const int par = (jt - jt0) & 1;
    __syncthreads();  // previous tile's LDS reads are done
    stage_write();
    if (tid == 0) flags[par ^ 1] = 0;
    __syncthreads();  // tile visible
    if (jt + 1 < jt1) stage_load(jt + 1);  // in flight under the MFMAs below

    // ---- S = K^T Q for this wave's 16 queries x 32 memory cells
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < kDe / 4; ++ks) {
      const float a0 = Kt[(4 * ks + g) * kKS + l15];
      const float a1 = Kt[(4 * ks + g) * kKS + 16 + l15];
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, qreg[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, qreg[ks], s1, 0, 0, 0);
    }
    // lane holds S[j = 4g + r (+16)][query l15]
    float sv[8];
    const int nbase = jt * kJT + 4 * g;
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sv[r] = (nbase + r < pl.M) ? s0[r] / a.sqrt_de : -INFINITY;          // models/rmnet.py:156
      sv[4 + r] = (nbase + 16 + r < pl.M) ? s1[r] / a.sqrt_de : -INFINITY;
      tmax = fmaxf(tmax, fmaxf(sv[r], sv[4 + r]));
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const bool bump = tmax > mref + kDefer;  // first tile: mref = -inf
    float alpha = 1.0f;
    if (bump) {
      alpha = __expf(mref - tmax);  // exp(-inf) = 0 on the first tile
      mref = tmax;
    }
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      sv[r] = expf(sv[r] - mref);
      rs += sv[r];
    }
    rs += __shfl_xor(rs, 16);
    rs += __shfl_xor(rs, 32);
    lsum = lsum * alpha + rs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Pt[(4 * g + r) * kPS + wave * 16 + l15] = sv[r];
      Pt[(16 + 4 * g + r) * kPS + wave * 16 + l15] = sv[4 + r];
    }
    if (g == 0) Al[wave * 16 + l15] = alpha;
    if (__any(bump) && lane == 0) flags[par] = 1;
    __syncthreads();  // P, alpha, flag visible

    // ---- O += V P for this wave's 128 value channels x 64 queries
    if (flags[par]) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float al = Al[it * 16 + l15];
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) acc[dt][it] *= al;
      }
    }
    const float* Vw = Vt + (wave * 128 + l15) * kVS + g;
#pragma unroll
    for (int ks = 0; ks < kJT / 4; ++ks) {
      float bq[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) bq[it] = Pt[(4 * ks + g) * kPS + it * 16 + l15];
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const float av = Vw[dt * 16 * kVS + 4 * ks];
#pragma unroll
        for (int it = 0; it < 4; ++it)
          acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[it], acc[dt][it], 0, 0, 0);
      }
    }
  }

  // ---- partial (O, m, l) -> workspace slot L in fragment order (common.h): 1 KB contiguous per store
  float* wo = a.ws_o + ((size_t)o * a.slots + L) * (size_t)kDo * kQT;
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
      *reinterpret_cast<f32x4*>(wo + partial_frag_offset(wave * 8 + dt, it, lane)) = acc[dt][it];
  if (g == 0) {
    float* wm = a.ws_ml + ((size_t)o * a.slots + L) * 2 * kQT;
    wm[wave * 16 + l15] = mref;
    wm[kQT + wave * 16 + l15] = lsum;
  }
}

#ifndef RMNET_COMB_ABL
#define RMNET_COMB_ABL 0   // experiments: 1 no q_val half, 2 no mem-half stores, 4 no fill blocks, 8 no partial loads
#endif
// Read-out channels (and as many q_val channels) per combine block: 16 for one or two objects (more,
// smaller workgroups: a launch is one latency chain deep), 32 beyond (fewer chains per byte moved).
// Measured at 480p/T=5: 1 object 20.0 vs 22.1 us, 8 objects 39.2 vs 33.1 us.
inline int comb_channels(int no) { return no <= 2 ? 16 : 32; }

// Merge the per-split partials of mr_main (fragment-ordered blocks, common.h).
// grid = (nqt_max + cell tiles, kDo / kCombCh, no).  The plan comes from the 32-byte record the read
// kernel left behind (one load instead of re-deriving it).
//   blocks [0, nqt_max)   : one compacted query tile x 64 channels each (exit beyond the live tiles):
//        4 lanes per query build the split weights w[s][q] = exp(m_s - m_tot) / l_tot (with the
//        closed-form N_out * exp(-m_tot) term of the masked memory cells); the 64 channels of a
//        split are ONE contiguous 16 KB run of fragments, read with 16-byte loads (a thread owns
//        the same query in all of them, so one weight per split), 16 loads in flight; the 64 x 64
//        tile goes through LDS.  The block then writes a CONTIGUOUS range of grid cells: its
//        queries' cells and the masked cells lying between them in raster order (those get the
//        mean-slot vector -- uniform soft-max, see file header -- and q_val * 0), so that every
//        output line has one writer; with them goes the q_val half of the cat (models/rmnet.py:163).
//   blocks [nqt_max, ...) : one tile of 64 grid cells each, for the rows ABOVE and BELOW the query
//        box only (mean-slot vector, q_val * 0).  Skipped when nothing is masked.
// Only mr_main's partials come here (natural-log running reference); the split-fp16 bank kernel merges its
// own (bank.hip) and this kernel returns at once when the drop-in entry's gate says the bank path ran.
// blocks [copy_x0, gridDim.x) (only when copy_x0 > 0): the q_val half of the cat as a streaming copy --
//        mem_val[o][Do + d][:] = q_val[o][d][:] * box, 16 bytes per lane along the channel rows (needs
//        h * w % 4 == 0, as on RMNet's grids).  The query-tile blocks used to carry it in 256-byte runs
//        per channel, which cost 14.5 of the kernel's 35 us at 8 objects; as blocks of their own the copy
//        overlaps the latency chain of the merge blocks.
template <bool REGIONAL, int kCombCh>
__global__ __launch_bounds__(kThreads, kCombCh == 16 ? 6 : 5) void mr_combine(const KArgs a, int nqt_max, int copy_x0) {   // (register cap: many resident workgroups hide the latency chain; LDS allows 5 at 32 channels)
  constexpr int kCombDt = kCombCh / 16;    // = channel tiles (fragments) per query tile and split
  __shared__ float Wt[kMaxSplits][kQT];
  __shared__ float Wm[kMaxSplits];
  __shared__ float red[4][kQT];
  __shared__ float Tt[kCombCh][kQT + 1];
  __shared__ float Tm[kCombCh];
  const int tid = threadIdx.x, o = blockIdx.z;
  if (a.gate && __builtin_amdgcn_readfirstlane(*a.gate) == 0) return;   // the split-fp16 bank path did the whole read
  Plan pl;
  int p8;
  {
    const int32_t* pr = a.ws_plan + (size_t)o * kPlanInts;
    pl.Mq = pr[0]; pl.nqt = pr[1]; pl.nsplit = pr[2]; pl.M = pr[3];
    pl.qr = Rect{pr[4], pr[5], pr[6], pr[7]};
    pl.njt = 0;
    p8 = __builtin_amdgcn_readfirstlane(pr[8]);
  }
  // Partial slots of one (object, query tile) pair (plan record, common.h): slot first + s * nqt + qt.
  struct PairSlots {
    int a0, sa, count;
    __device__ inline int slot(int s_) const { return a0 + s_ * sa; }
  };
  auto pair_slots = [&](int qt_) -> PairSlots { return PairSlots{p8 + qt_, pl.nqt, pl.nsplit}; };
  if (copy_x0 > 0 && (int)blockIdx.x >= copy_x0) {
    const int hw4 = a.hw >> 2, n4 = kCombCh * hw4, d0c = blockIdx.y * kCombCh;
    const Rect rc = REGIONAL ? pl.qr : Rect{0, a.w - 1, 0, a.h - 1};
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.qv + ((size_t)o * kDo + d0c) * a.hw);
    f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(a.out + ((size_t)o * 2 * kDo + kDo + d0c) * a.hw);
    const int stride = ((int)gridDim.x - copy_x0) * kThreads;
    for (int i = ((int)blockIdx.x - copy_x0) * kThreads + tid; i < n4; i += stride) {
      f32x4 v = src[i];                                   // (rows of a channel group are contiguous: index = d * hw4 + c4)
      const int d = i / hw4, cell = (i - d * hw4) << 2;
      int cy = cell / a.w, cx = cell - cy * a.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = rc.contains(cy, cx) ? v[e] : v[e] * 0.0f;  // q_val * box (:358), x*0 semantics
        if (++cx == a.w) { cx = 0; ++cy; }
      }
      dst[i] = v;
    }
    return;
  }
  const bool qv_here = copy_x0 == 0;                      // the merge / fill blocks below write the q_val half themselves
  const float vun = 1.0f;
  const int qi = tid & 63, sl = tid >> 6;
  const float n_out = (float)(a.T * a.hw - pl.M);
  const float* __restrict__ ml = a.ws_ml;
  const float* __restrict__ wo = a.ws_o;
  const bool fill = (int)blockIdx.x >= nqt_max;   // masked-row filler block
  const bool masked = REGIONAL && pl.Mq < a.hw;   // some query cell is masked: a mean slot exists
  const int qt = (int)blockIdx.x;
  if (!fill && qt >= pl.nqt) return;
  if (fill && (!masked || (RMNET_COMB_ABL & 4))) return;
  const int n0 = qt * kQT, n1 = min(n0 + kQT, pl.Mq);   // this tile's real queries
  if (!fill && n1 <= n0) return;                        // the tile holds only the mean slot
  int fc0 = 0;
  if (fill) {   // cell tile: only rows outside the box's row range are written here
    fc0 = ((int)blockIdx.x - nqt_max) * kQT;
    const int y0 = fc0 / a.w, y1 = (min(fc0 + kQT, a.hw) - 1) / a.w;
    if (pl.Mq > 0 && y0 >= pl.qr.cy0 && y1 <= pl.qr.cy1) return;
  }
  const int d0 = blockIdx.y * kCombCh;
  const PairSlots ps = fill ? PairSlots{0, 1, 0} : pair_slots(qt);
  const int nsp = ps.count;                             // partials of this query tile
  constexpr size_t kSlotF = (size_t)kDo * kQT;          // floats per partial slot
  auto ex = [&](float x) { return expf(x); };

  // First batch of this tile's partial fragments: requested NOW, before the weights are known, so
  // that their latency overlaps the (m, l) loads and the three barriers of the weight phase.
  // Thread = (query tile it = sl, lane qi) of the block's channel tiles; splits past the last one
  // are clamped (re-read) and get weight 0.
  constexpr int kU = kCombDt >= 8 ? 1 : 8 / kCombDt;   // splits per batch: 8 independent 16-byte loads in flight
  constexpr int kE = kU < 4 ? kU : 4;    // splits of the early batch (small: duplicates cost bandwidth)
  const float* __restrict__ psrc = wo + partial_frag_offset(d0 >> 4, sl, qi);
  f32x4 v0[kE][kCombDt];
#pragma unroll
  for (int u = 0; u < kE; ++u)
#pragma unroll
    for (int k = 0; k < kCombDt; ++k) v0[u][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!fill && nsp > 0 && !(RMNET_COMB_ABL & 8)) {   // (no split at all when every memory cell is masked: the read-out is 0)
#pragma unroll
    for (int u = 0; u < kE; ++u)
#pragma unroll
      for (int k = 0; k < kCombDt; ++k)
        v0[u][k] = *reinterpret_cast<const f32x4*>(psrc + (size_t)ps.slot(min(u, nsp - 1)) * kSlotF + (size_t)k * 1024);
  }

  // Everything the weights need is requested before the first barrier: this tile's (m, l) pairs of up
  // to 16 splits in registers (the rest, rare, is re-read in both passes below), the mean slot's
  // pairs in wave 0.  Three barriers in all; the mean-slot channels are fetched while the weights
  // are being formed.
  constexpr int kMl = 4;                  // register-held splits per thread (x 4 thread groups = 16)
  constexpr int kTmParts = kThreads / kCombCh;
  __shared__ float red2[4][kQT];
  __shared__ float tp[kTmParts][kCombCh];
  const int qtm = pl.Mq >> 6, mq = pl.Mq & 63;
  const PairSlots pm = masked ? pair_slots(qtm) : PairSlots{0, 1, 0};   // the mean slot's query tile
  float m_r[kMl], l_r[kMl];
  if (!fill) {
#pragma unroll
    for (int j = 0; j < kMl; ++j) {
      const int sj = sl + 4 * j;
      const float* e = ml + ((size_t)ps.slot(min(sj, max(nsp - 1, 0))) * 2) * kQT;
      const bool on = sj < nsp;
      m_r[j] = on ? e[qi] : -INFINITY;
      l_r[j] = on ? e[kQT + qi] : 0.0f;
    }
  }
  float mm = -INFINITY, lm = 0.0f;        // mean slot (wave 0, lane = split)
  if (masked && tid < RMNET_WAVE && tid < pm.count) {
    const float* e = ml + ((size_t)pm.slot(tid) * 2) * kQT;
    mm = e[mq];
    lm = e[kQT + mq];
  }

  // ---- pass 1: row maxima; wave 0 also finishes the mean slot's weights (shuffles only)
  if (!fill) {
    float mloc = -INFINITY;
#pragma unroll
    for (int j = 0; j < kMl; ++j) mloc = fmaxf(mloc, m_r[j]);
    for (int sj = sl + 4 * kMl; sj < nsp; sj += 4) mloc = fmaxf(mloc, ml[((size_t)ps.slot(sj) * 2) * kQT + qi]);
    red[sl][qi] = mloc;
  }
  if (masked && tid < RMNET_WAVE) {
    float mtot = mm;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mtot = fmaxf(mtot, __shfl_xor(mtot, d));
    if (n_out > 0.0f) mtot = fmaxf(mtot, 0.0f);
    const float wgt = tid < pm.count ? ex(mm - mtot) : 0.0f;
    float ltot = lm * wgt;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ltot += __shfl_xor(ltot, d);
    if (n_out > 0.0f) ltot += n_out * ex(-mtot);
    Wm[tid] = wgt / ltot * vun;
  }
  __syncthreads();                                                         // barrier 1

  // ---- mean-slot channels (requested now, summed after the weights) + pass 2: the weights
  float tm_acc = 0.0f;
  if (masked) {   // 256 threads = kCombCh channels x kTmParts interleaved subsets of the splits
    const int chn = tid % kCombCh, part = tid / kCombCh;
    const float* __restrict__ src = wo + partial_elem_offset(mq, d0 + chn);
#pragma unroll 4
    for (int s2 = part; s2 < pm.count; s2 += kTmParts) tm_acc += Wm[s2] * src[(size_t)pm.slot(s2) * kSlotF];
  }
  if (!fill) {
    float mtot = fmaxf(fmaxf(red[0][qi], red[1][qi]), fmaxf(red[2][qi], red[3][qi]));
    if (n_out > 0.0f) mtot = fmaxf(mtot, 0.0f);        // the N_out masked memory cells have S = 0
    float lloc = 0.0f;
#pragma unroll
    for (int j = 0; j < kMl; ++j) {
      const int sj = sl + 4 * j;
      const float wgt = ex(m_r[j] - mtot);              // (-inf for a split that does not exist: 0)
      if (sj < kMaxSplits) Wt[sj][qi] = wgt;
      lloc += l_r[j] * wgt;
    }
    for (int sj = sl + 4 * kMl; sj < nsp; sj += 4) {
      const float* e = ml + ((size_t)ps.slot(sj) * 2) * kQT;
      const float wgt = ex(e[qi] - mtot);
      Wt[sj][qi] = wgt;
      lloc += e[kQT + qi] * wgt;
    }
    // (the closed-form term of the N_out masked memory cells rides in group 0's partial sum)
    red2[sl][qi] = lloc + (sl == 0 && n_out > 0.0f ? n_out * ex(-mtot) : 0.0f);
  }
  if (masked) tp[tid / kCombCh][tid % kCombCh] = tm_acc;
  __syncthreads();                                                         // barrier 2
  if (masked && tid < kCombCh) {
    float t = 0.0f;
#pragma unroll
    for (int pp = 0; pp < kTmParts; ++pp) t += tp[pp][tid];
    Tm[tid] = t;
  }

  if (fill) {
    __syncthreads();                                                       // (Tm visible)
    const int cell = fc0 + qi;
    if (cell >= a.hw) return;
    const int cy = cell / a.w;
    if (pl.Mq > 0 && cy >= pl.qr.cy0 && cy <= pl.qr.cy1) return;   // written by the query-tile blocks
    float* __restrict__ out = a.out + (size_t)o * 2 * kDo * a.hw + cell;
    const float* __restrict__ qv = a.qv + (size_t)o * kDo * a.hw + cell;
#pragma unroll 8
    for (int dd = sl; dd < kCombCh; dd += 4) {
      const int d = d0 + dd;
      if (!(RMNET_COMB_ABL & 2)) out[(size_t)d * a.hw] = Tm[dd];
      if (!(RMNET_COMB_ABL & 1) && qv_here) out[(size_t)(kDo + d) * a.hw] = qv[(size_t)d * a.hw] * 0.0f;   // q_val * box (:358), x*0 semantics
    }
    return;
  }

  {  // accumulate: thread = (query tile it = sl, lane qi) of the channel tiles of this block
    const int q = sl * 16 + (qi & 15), gg = qi >> 4;
    const float inv = vun / (red2[0][q] + red2[1][q] + red2[2][q] + red2[3][q]);
    f32x4 acc[kCombDt];
#pragma unroll
    for (int k = 0; k < kCombDt; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ src = psrc;
#pragma unroll
    for (int u = 0; u < kE; ++u) {
      const float wgt = u < nsp ? Wt[u][q] : 0.0f;   // (clamped duplicates of the early batch)
#pragma unroll
      for (int k = 0; k < kCombDt; ++k) acc[k] += wgt * v0[u][k];
    }
    int s = kE;
    for (; s + kU <= nsp; s += kU) {
      f32x4 v[kU][kCombDt];
#pragma unroll
      for (int u = 0; u < kU; ++u)
#pragma unroll
        for (int k = 0; k < kCombDt; ++k)
          v[u][k] = *reinterpret_cast<const f32x4*>(src + (size_t)ps.slot(s + u) * kSlotF + (size_t)k * 1024);
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const float wgt = Wt[s + u][q];
#pragma unroll
        for (int k = 0; k < kCombDt; ++k) acc[k] += wgt * v[u][k];
      }
    }
    for (; s < nsp; ++s) {
      const float wgt = Wt[s][q];
#pragma unroll
      for (int k = 0; k < kCombDt; ++k)
        acc[k] += wgt * *reinterpret_cast<const f32x4*>(src + (size_t)ps.slot(s) * kSlotF + (size_t)k * 1024);
    }
#pragma unroll
    for (int k = 0; k < kCombDt; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) Tt[k * 16 + 4 * gg + r][q] = acc[k][r] * inv;
  }
  __syncthreads();                                                         // barrier 3

  // ---- write-out: the contiguous cell range this tile owns
  float* __restrict__ outo = a.out + (size_t)o * 2 * kDo * a.hw;
  const float* __restrict__ qvo = a.qv + (size_t)o * kDo * a.hw;
  if (!masked) {   // every cell is a query: tile qt = cells [n0, n1)
    const int cell = n0 + qi;
    if (cell >= n1) return;
#pragma unroll 8
    for (int dd = sl; dd < kCombCh; dd += 4) {
      const int d = d0 + dd;
      outo[(size_t)d * a.hw + cell] = Tt[dd][qi];
      if (qv_here) outo[(size_t)(kDo + d) * a.hw + cell] = qvo[(size_t)d * a.hw + cell];     // cat(mem, q_val), :163
    }
    return;
  }
  const int rw = pl.qr.width();
  const int c_begin = n0 == 0 ? pl.qr.cy0 * a.w : query_cell(pl, a.w, n0);
  const int c_end = n1 == pl.Mq ? (pl.qr.cy1 + 1) * a.w : query_cell(pl, a.w, n1);   // exclusive
  for (int cell = c_begin + qi; cell < c_end; cell += kQT) {
    const int cy = cell / a.w, cx = cell - cy * a.w;
    const bool inside = cx >= pl.qr.cx0 && cx <= pl.qr.cx1;   // (the row is inside by construction)
    const int q = inside ? (cy - pl.qr.cy0) * rw + (cx - pl.qr.cx0) - n0 : 0;
#pragma unroll 8
    for (int dd = sl; dd < kCombCh; dd += 4) {
      const int d = d0 + dd;
      if (!(RMNET_COMB_ABL & 2)) outo[(size_t)d * a.hw + cell] = inside ? Tt[dd][q] : Tm[dd];
      if (!(RMNET_COMB_ABL & 1) && qv_here) {
        const float x = qvo[(size_t)d * a.hw + cell];
        outo[(size_t)(kDo + d) * a.hw + cell] = inside ? x : x * 0.0f;   // q_val * box (:358), x*0 semantics
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Generic path (any De <= 256, any Do) and the optional p output: a plain three-pass soft-max
// that materialises p, then a plain V p product.  Correct for every shape, not tuned: RMNet only
// ever uses De = 128 / Do = 512 and never reads p (models/rmnet.py:361-366).
// ------------------------------------------------------------------------------------------
struct GArgs {
  const float *mk, *mv, *qk, *qv;
  float* out;
  float* p;  // [no][THW][hw]
  const int32_t* mem_rects;
  const int32_t* qry_rects;
  int no, De, Do, T, h, w, hw;
  long long mk_cs, mk_os, mv_cs, mv_os;
  float sqrt_de;
};

__device__ inline bool mem_live(const GArgs& a, int o, int j) {
  if (!a.mem_rects) return true;
  const int t = j / a.hw, pos = j - t * a.hw;
  const int cy = pos / a.w, cx = pos - cy * a.w;
  const int32_t* r = a.mem_rects + ((size_t)o * a.T + t) * 4;
  return cx >= r[0] && cx <= r[1] && cy >= r[2] && cy <= r[3];
}
__device__ inline bool qry_live(const GArgs& a, int o, int cell) {
  if (!a.qry_rects) return true;
  const int cy = cell / a.w, cx = cell - cy * a.w;
  const int32_t* r = a.qry_rects + (size_t)o * 4;
  return cx >= r[0] && cx <= r[1] && cy >= r[2] && cy <= r[3];
}

__global__ __launch_bounds__(kThreads) void p_kernel(const GArgs a) {
  extern __shared__ float qs[];  // [De][64] query tile, then red[4][64]
  float* red = qs + a.De * 64;
  const int tid = threadIdx.x, ci = tid & 63, jj = tid >> 6, o = blockIdx.y;
  const int cell = blockIdx.x * 64 + ci;
  const bool live = cell < a.hw;
  const bool qin = live && qry_live(a, o, cell);
  for (int c = jj; c < a.De; c += 4)
    qs[c * 64 + ci] = qin ? a.qk[(size_t)o * a.De * a.hw + (size_t)c * a.hw + cell] : 0.0f;
  __syncthreads();
  const int thw = a.T * a.hw;
  float* pc = a.p + (size_t)o * thw * a.hw + cell;
  const float* mk = a.mk + (size_t)o * a.mk_os;
  float mx = -INFINITY;
  for (int j = jj; j < thw; j += 4) {
    float dot = 0.0f;
    if (mem_live(a, o, j))
      for (int c = 0; c < a.De; ++c) dot += mk[(size_t)c * a.mk_cs + j] * qs[c * 64 + ci];
    const float sc = dot / a.sqrt_de;
    if (live) pc[(size_t)j * a.hw] = sc;
    mx = fmaxf(mx, sc);
  }
  red[jj * 64 + ci] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[ci], red[64 + ci]), fmaxf(red[128 + ci], red[192 + ci]));
  __syncthreads();
  float sum = 0.0f;
  if (live)
    for (int j = jj; j < thw; j += 4) {
      const float e = expf(pc[(size_t)j * a.hw] - mx);
      pc[(size_t)j * a.hw] = e;
      sum += e;
    }
  red[jj * 64 + ci] = sum;
  __syncthreads();
  sum = red[ci] + red[64 + ci] + red[128 + ci] + red[192 + ci];
  if (live)
    for (int j = jj; j < thw; j += 4) pc[(size_t)j * a.hw] = pc[(size_t)j * a.hw] / sum;
}

__global__ __launch_bounds__(kThreads) void pv_kernel(const GArgs a) {
  const int tid = threadIdx.x, ci = tid & 63, dg = tid >> 6, o = blockIdx.z;
  const int cell = blockIdx.x * 64 + ci;
  if (cell >= a.hw) return;
  const int thw = a.T * a.hw;
  const int d0 = (blockIdx.y * 4 + dg) * 4;
  const float* pc = a.p + (size_t)o * thw * a.hw + cell;
  const float* mv = a.mv + (size_t)o * a.mv_os;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < thw; ++j) {
    if (!mem_live(a, o, j)) continue;
    const float pj = pc[(size_t)j * a.hw];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (d0 + u < a.Do) acc[u] += mv[(size_t)(d0 + u) * a.mv_cs + j] * pj;
  }
  const bool qin = qry_live(a, o, cell);
  float* out = a.out + (size_t)o * 2 * a.Do * a.hw + cell;
  const float* qv = a.qv + (size_t)o * a.Do * a.hw + cell;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (d0 + u < a.Do) {
      out[(size_t)(d0 + u) * a.hw] = acc[u];
      const float v = qv[(size_t)(d0 + u) * a.hw];
      out[(size_t)(a.Do + d0 + u) * a.hw] = qin ? v : v * 0.0f;
    }
}

inline bool fast_shape(int De, int Do, int T, int flags) {
  return De == kDe && Do == kDo && T <= kMaxT && !(flags & RMNET_MR_FORCE_GENERIC);
}

#ifndef RMNET_COMB_COPY_BLOCKS
#define RMNET_COMB_COPY_BLOCKS 8   // copy blocks per (object, channel group); measured at 8 objects: 1: 46 us, 4: 35, 8: 30-31, 16-32: 32 (39 without)
#endif
inline void launch_combine(bool regional, int cch, dim3 grid, hipStream_t st, const KArgs& a, int nqt_max) {
  // the q_val half as copy blocks of their own when the rows can be moved 16 bytes at a time
  int copy_x0 = 0;
  if (a.hw % 4 == 0 && (reinterpret_cast<uintptr_t>(a.qv) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
      !(RMNET_COMB_ABL & 1)) {
    copy_x0 = (int)grid.x;
    grid.x += RMNET_COMB_COPY_BLOCKS;
  }
  if (regional) {
    if (cch == 16) hipLaunchKernelGGL((mr_combine<true, 16>), grid, dim3(kThreads), 0, st, a, nqt_max, copy_x0);
    else hipLaunchKernelGGL((mr_combine<true, 32>), grid, dim3(kThreads), 0, st, a, nqt_max, copy_x0);
  } else {
    if (cch == 16) hipLaunchKernelGGL((mr_combine<false, 16>), grid, dim3(kThreads), 0, st, a, nqt_max, copy_x0);
    else hipLaunchKernelGGL((mr_combine<false, 32>), grid, dim3(kThreads), 0, st, a, nqt_max, copy_x0);
  }
}

inline int slots_for(int no, int hw) {
  const int nqt_max = (hw + 1 + kQT - 1) / kQT;
  const int per_obj = (kTargetSlots + no - 1) / no;
  return nqt_max > per_obj ? nqt_max : per_obj;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Partial slots of the drop-in entry's workspace: enough for mr_main's per-object grid AND for bk_main's
// launch-wide chunks (the two share one partial area; only one of them writes it in a given call).
inline size_t mr_slots(int no, int hw) {
  const size_t a = (size_t)no * slots_for(no, hw), b = (size_t)bank_total_slots(no, hw);
  return a > b ? a : b;
}

}  // namespace

size_t memory_read_ws_bytes(int no, int De, int Do, int T, int h, int w, int flags) {
  const size_t hw = (size_t)h * w;
  if (fast_shape(De, Do, T, flags)) {
    const size_t slots = mr_slots(no, (int)hw);
    return align256(slots * kDo * kQT * 4) + align256(slots * 2 * kQT * 4) + align256((size_t)no * kPlanInts * 4) +
           ((flags & RMNET_MR_EXACT_FP32) ? 0 : align256(bank_bytes(no, T, h, w)));   // + the transient bank
  }
  return align256((size_t)no * T * hw * hw * 4);  // a p-sized buffer
}

int launch_memory_read(const MemReadArgs& m, hipStream_t st) {
  if (!m.mk || !m.mv || !m.qk || !m.qv || !m.out) return RMNET_E_INVALID_ARG;
  if (m.no <= 0 || m.De <= 0 || m.Do <= 0 || m.T <= 0 || m.h <= 0 || m.w <= 0)
    return RMNET_E_INVALID_ARG;
  if ((m.mem_rects == nullptr) != (m.qry_rects == nullptr)) return RMNET_E_INVALID_ARG;
  const long long hw = (long long)m.h * m.w, thw = hw * m.T;
  if (thw > (1LL << 30) || m.no > 65535) return RMNET_E_UNSUPPORTED;
  const long long mk_cs = m.mk_cs ? m.mk_cs : thw, mv_cs = m.mv_cs ? m.mv_cs : thw;
  const long long mk_os = m.mk_os ? m.mk_os : mk_cs * m.De, mv_os = m.mv_os ? m.mv_os : mv_cs * m.Do;
  if (mk_cs < thw || mv_cs < thw) return RMNET_E_INVALID_ARG;
  const bool regional = m.mem_rects != nullptr;
  const bool fast = fast_shape(m.De, m.Do, m.T, m.flags);
  const float sqrt_de = sqrtf((float)m.De);

  if (fast) {
    if (!m.ws || m.ws_bytes < memory_read_ws_bytes(m.no, m.De, m.Do, m.T, m.h, m.w, m.flags))
      return RMNET_E_WORKSPACE;
    // Default: stage the memory into a TRANSIENT split-fp16 bank (one launch over all T frames, same
    // bytes as the fp32 source) and run the bank read -- 3-4x faster than the exact-fp32 MFMA kernel
    // even with the staging pass.  The staging kernel counts elements outside fp16's window; if there
    // are any, bk_main returns at once and mr_main (exact fp32, no range limit) does the read instead --
    // decided on the device, no host sync.  RMNET_MR_EXACT_FP32 forces mr_main.
    const bool via_bank = !(m.flags & RMNET_MR_EXACT_FP32) && (long long)m.no * m.T <= 65535;
    const size_t nslots = mr_slots(m.no, (int)hw);
    KArgs a;
    a.mk = m.mk; a.mv = m.mv; a.qk = m.qk; a.qv = m.qv; a.out = m.out;
    a.mem_rects = m.mem_rects; a.qry_rects = m.qry_rects;
    a.no = m.no; a.T = m.T; a.h = m.h; a.w = m.w; a.hw = (int)hw;
    a.mk_cs = mk_cs; a.mk_os = mk_os; a.mv_cs = mv_cs; a.mv_os = mv_os;
    a.slots = slots_for(m.no, (int)hw);
    a.sqrt_de = sqrt_de;
    a.gate = nullptr;
    a.ws_o = static_cast<float*>(m.ws);
    a.ws_ml = reinterpret_cast<float*>(static_cast<char*>(m.ws) + align256(nslots * kDo * kQT * 4));
    a.ws_plan = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a.ws_ml) + align256(nslots * 2 * kQT * 4));
    char* bank = reinterpret_cast<char*>(a.ws_plan) + align256((size_t)m.no * kPlanInts * 4);
    // (the optional timing events bracket the WHOLE call: staging pass included)
    if (m.ev_start && hipEventRecord(m.ev_start, st) != hipSuccess) return RMNET_E_LAUNCH;
    if (via_bank) {
      const BankView b = bank_view(bank, m.no, m.T, m.h, m.w);
      // control block of the transient bank: the overflow word and the pairs' arrival tickets (the workspace is arbitrary memory)
      if (int e = launch_bank_ctl_clear(b.ovf, (int)(bank_ctl_bytes(m.no, m.h, m.w) / 4), st)) return e;
      if (int e = launch_bank_stage(bank, m.no, m.T, m.h, m.w, 0, m.T, m.mk, m.mv, mk_cs, mk_os, mv_cs, mv_os,
                                    m.mem_rects, st, nullptr, /*colsum=*/regional))
        return e;
      a.gate = b.ovf;
    }
    const int nqt_max = (int)((hw + 1 + kQT - 1) / kQT);
    dim3 g1(a.slots, m.no);
    const int cch = comb_channels(m.no);
    dim3 g2((unsigned)(nqt_max + (regional ? (hw + kQT - 1) / kQT : 0)), kDo / cch, m.no);
    if (via_bank) {
      BankReadArgs r;
      r.bank = bank; r.no = m.no; r.Tcap = m.T; r.h = m.h; r.w = m.w; r.T = m.T;
      r.qk = m.qk; r.qv = m.qv; r.qry_rects = m.qry_rects; r.out = m.out;
      r.ws_o = a.ws_o; r.ws_ml = a.ws_ml; r.ws_plan = a.ws_plan; r.slots = (int)nslots;
      r.ws = nullptr; r.ws_bytes = 0; r.gate = 1;
      r.f16 = (m.flags & RMNET_MR_F16) ? 1 : (m.flags & RMNET_MR_QX) ? 2 : 0;
      if (int e = launch_bank_main(r, st)) return e;
    }
    // (mr_main plans its own ceil(M / 32) compacted tiles from the rectangles: the transient bank's per-frame tile
    //  count must never enter its plan -- round 2 leaked it there and small boxes produced NaN)
    if (regional)
      hipLaunchKernelGGL(mr_main<true>, g1, dim3(kThreads), 0, st, a);
    else
      hipLaunchKernelGGL(mr_main<false>, g1, dim3(kThreads), 0, st, a);
    if (int e = check_launch()) return e;
    if (m.ev_mid && hipEventRecord(m.ev_mid, st) != hipSuccess) return RMNET_E_LAUNCH;
    if (regional)
      launch_combine(true, cch, g2, st, a, nqt_max);
    else
      launch_combine(false, cch, g2, st, a, nqt_max);
    if (int e = check_launch()) return e;
    if (m.ev_end && hipEventRecord(m.ev_end, st) != hipSuccess) return RMNET_E_LAUNCH;
    if (!m.p_out) return RMNET_OK;
  }

  // generic path and/or the p output
  if (m.De > 224) return RMNET_E_UNSUPPORTED;  // query tile must fit 64 KB of dynamic LDS
  GArgs ga;
  ga.mk = m.mk; ga.mv = m.mv; ga.qk = m.qk; ga.qv = m.qv; ga.out = m.out;
  ga.mem_rects = m.mem_rects; ga.qry_rects = m.qry_rects;
  ga.no = m.no; ga.De = m.De; ga.Do = m.Do; ga.T = m.T; ga.h = m.h; ga.w = m.w; ga.hw = (int)hw;
  ga.mk_cs = mk_cs; ga.mk_os = mk_os; ga.mv_cs = mv_cs; ga.mv_os = mv_os;
  ga.sqrt_de = sqrt_de;
  ga.p = m.p_out;
  if (!ga.p) {
    if (!m.ws || m.ws_bytes < align256((size_t)m.no * thw * hw * 4)) return RMNET_E_WORKSPACE;
    ga.p = static_cast<float*>(m.ws);
  }
  const size_t shm = ((size_t)m.De * 64 + 256) * sizeof(float);
  hipLaunchKernelGGL(p_kernel, dim3((unsigned)((hw + 63) / 64), m.no), dim3(kThreads), shm, st, ga);
  if (int e = check_launch()) return e;
  if (!fast) {
    hipLaunchKernelGGL(pv_kernel, dim3((unsigned)((hw + 63) / 64), (m.Do + 15) / 16, m.no),
                       dim3(kThreads), 0, st, ga);
    if (int e = check_launch()) return e;
  }
  return RMNET_OK;
}

size_t bank_read_ws_bytes(int no, int h, int w) {
  const size_t slots = bank_total_slots(no, h * w);
  return align256(slots * kDo * kQT * 4) + align256(slots * 2 * kQT * 4) + align256((size_t)no * kPlanInts * 4);
}

size_t bank_read_ws_bytes_T(int no, int h, int w, int T) {
  const size_t one = bank_read_ws_bytes(no, h, w);
  const int per = bank_max_frames_per_launch();
  if (T <= per) return one;
  const size_t nchunk = ((size_t)T + per - 1) / per, hw = (size_t)h * w;
  return one + align256((nchunk - 1) * no * 2 * kDo * hw * 4) + align256(nchunk * no * 2 * hw * 4);
}

static int bank_read_one(BankReadArgs& m, hipStream_t st) {
  const int hw = m.h * m.w;
  const size_t tslots = bank_total_slots(m.no, hw);
  m.slots = (int)tslots;
  m.ws_o = static_cast<float*>(m.ws);
  m.ws_ml = reinterpret_cast<float*>(static_cast<char*>(m.ws) + align256(tslots * kDo * kQT * 4));
  m.ws_plan = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(m.ws_ml) + align256(tslots * 2 * kQT * 4));
  return launch_bank_main(m, st);
}

int launch_bank_read(BankReadArgs& m, hipStream_t st) {
  if (!m.bank || !m.qk || !m.qv || !m.out) return RMNET_E_INVALID_ARG;
  if (m.no <= 0 || m.Tcap <= 0 || m.h <= 0 || m.w <= 0 || m.T > m.Tcap || (m.T <= 0 && !m.T_dev) || m.T < 0)
    return RMNET_E_INVALID_ARG;
  if (m.no > 65535) return RMNET_E_UNSUPPORTED;
  const int per = bank_max_frames_per_launch();
  // a frame count that lives on the device cannot be chunked by the host: such banks hold at most one launch's frames
  if (m.T_dev && m.Tcap > per) return RMNET_E_UNSUPPORTED;
  if (!m.ws || m.ws_bytes < bank_read_ws_bytes_T(m.no, m.h, m.w, m.T)) return RMNET_E_WORKSPACE;
  // The read kernel's queue words and arrival counters must be zero when it starts.  Every read leaves them so, but an
  // aborted launch (or a merge that timed out) would poison every later read of the bank: clear them per call
  // (cdna_hip_programming.md section 6 Guideline 16, "re-initialise every call"; a small kernel: launch_bank_ctl_clear).  The first
  // 64 bytes (overflow / time-out words) are the bank's own sticky state and stay.
  const BankView b = bank_view(m.bank, m.no, m.Tcap, m.h, m.w);
#ifndef RMNET_NO_CTL_CLEAR   // (experiments only)
  if (int e = launch_bank_ctl_clear(b.ovf + 16, (int)(bank_ctl_bytes(m.no, m.h, m.w) / 4) - 16, st)) return e;
#endif
  // ONE kernel: bk_main writes the q_val half and the masked cells while it waits for its plan, reads, and the last
  // arriver of every (object, query tile) pair merges the pair's partials and writes the read-out (bank.hip).
  // ev_mid is kept for the callers that bracket "main" and "combine" separately: the second bracket is now empty.
  if (m.ev_start && hipEventRecord(m.ev_start, st) != hipSuccess) return RMNET_E_LAUNCH;
  if (m.T <= per) {
    if (int e = bank_read_one(m, st)) return e;
  } else {
    // more memorised frames than one launch takes (LDS prefix arrays): chunks of `per` slots, merged by bk_chain (bank.hip)
    const int nchunk = (m.T + per - 1) / per, hw = m.h * m.w;
    if (nchunk > bank_chain_max_chunks()) return RMNET_E_UNSUPPORTED;
    char* extra = static_cast<char*>(m.ws) + bank_read_ws_bytes(m.no, m.h, m.w);
    float* tmp = reinterpret_cast<float*>(extra);
    float* ml = reinterpret_cast<float*>(extra + align256((size_t)(nchunk - 1) * m.no * 2 * kDo * hw * 4));
    float* const out = m.out;
    const int T = m.T;
    for (int c = 0; c < nchunk; ++c) {
      BankReadArgs r = m;
      r.ev_start = r.ev_mid = r.ev_end = nullptr;
      r.t0 = c * per;
      r.T = T - r.t0 < per ? T - r.t0 : per;
      r.out = c == 0 ? out : tmp + (size_t)(c - 1) * m.no * 2 * kDo * hw;
      r.ml_out = ml + (size_t)c * m.no * 2 * hw;
      if (int e = launch_bank_ml_fill(r.ml_out, m.no, hw, (float)r.T * (float)hw, st)) return e;
      if (c > 0)
        if (int e = launch_bank_ctl_clear(b.ovf + 16, (int)(bank_ctl_bytes(m.no, m.h, m.w) / 4) - 16, st)) return e;
      if (int e = bank_read_one(r, st)) return e;
    }
    if (int e = launch_bank_chain(out, tmp, ml, m.no, hw, nchunk, st)) return e;
  }
  if (m.ev_mid && hipEventRecord(m.ev_mid, st) != hipSuccess) return RMNET_E_LAUNCH;
  if (m.ev_end && hipEventRecord(m.ev_end, st) != hipSuccess) return RMNET_E_LAUNCH;
  return RMNET_OK;
}

}  // namespace rmnet
