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
//             compacts to the box, splits, transposes 128-channel chunks through LDS (K to
//             cell-major rows, V to fragment order), writes the slot.
// bk_main   : the read.  Workgroup = 12 waves = 64 compacted queries x one split of the tile list
//             (the splits of all objects of a launch are planned together on the device).
//             4 producer waves compute S for 16 queries each (v_mfma_f32_16x16x32_f16, K tile shared
//             through a swizzled LDS ring), do the online soft-max in registers (a query's row
//             lives in 4 lanes x 8 registers), split P to fp16 hi/lo in the MFMA B-fragment layout
//             and publish the fragments in LDS; 8 consumer waves accumulate O += V P for 64 value
//             channels x all 64 queries each, with the V A-fragments loaded straight from the bank
//             into registers (16 B / lane, no LDS, no transpose) one tile ahead, and feed the K
//             ring.  Split mode: ONE barrier per 32-cell tile, all waves in step.  fp16-operand modes [r6]:
//             a step is 64 cells and TWO barrier intervals, the two consumers of a SIMD work half a step
//             apart (one multiplies while the other loads) -- see the PING-PONG section below.
//             The SAME launch also does everything else of MemoryReader.forward: while a workgroup waits for
//             the first dependent loads of its plan it streams its share of the soft-max-independent
//             outputs -- the q_val half of the cat (x box) and the read-out of MASKED query cells (= the
//             mean of m_val over all T*h*w cells, from the per-slot column sums bk_append leaves in the
//             bank) -- and when a workgroup finishes a segment it publishes its partial (O, m, l) with
//             write-through stores and draws a ticket; the last arriver of an (object, query tile) pair
//             merges the pair's partials, adds the closed-form term of the masked memory cells,
//             normalises and scatters the read-out to the query cells.  No combine kernel, no second pass
//             over the partials by another launch.
// bk_colsum : finishes a slot's column sums of the values (fixed summation order: reads are repeatable).
#include <type_traits>

#include "common.h"

namespace rmnet {
namespace {

constexpr int kDe = 128, kDo = 512;
constexpr int kQT = 64, kJT = 32;
constexpr int kThreads = 256;
constexpr int kMaxT = 2048;   // memorised frames per read (LDS: tile prefix + cells per frame, 16 KB)
constexpr float kDeferLog2 = 11.5415603f;   // 8 * log2(e): P <= e^8 stays far inside fp16 range (65504)
// The S accumulators are in RAW units: keys are stored times 2^6 and the query fragments carry
// log2(e) / sqrt(De) times 2^6 as well (both operands then sit where fp16's hi AND lo planes are
// normal numbers), so S_log2 = kSraw * S_raw with kSraw = 2^-12; the factor rides in the soft-max's
// fma (exp2(S_raw * kSraw - m)), it costs no instruction.
constexpr float kSraw = 1.0f / 4096.0f;
constexpr float kDeferRaw = kDeferLog2 * 4096.0f;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2 __attribute__((ext_vector_type(2)));

__device__ inline void split_f16(float x, _Float16& hi, _Float16& lo) {
  const float c = fminf(fmaxf(x, -65504.0f), 65504.0f);   // saturate instead of inf/NaN
  hi = (_Float16)c;
  lo = (_Float16)fminf(fmaxf(x - (float)hi, -65504.0f), 65504.0f);
}
// Keys and values are stored times 2^kBankShift (exact): fp16's window then covers |x| from
// 65504 / 64 = 1023.5 down to an absolute error floor of 2^-25 / 64 = 4.7e-10 per element (the lo
// plane is subnormal below |x| = 2^-3 / 64), which is where conv-net activations live; 1/64 is folded
// back into the query scale (keys) and into the combine's normalisation (values).  An element beyond
// +-1023.5 saturates and is COUNTED in the bank's overflow word (rmnet_bank_overflow_count).
constexpr float kBankScale = 64.0f;      // 2^6
// Sticky error bits of the overflow word (any non-zero value = "do not trust this bank's reads"):
constexpr int kBankBadSlot = 1 << 30;    // an append / read with a slot or frame count outside [0, Tcap] (device counter out of step)
constexpr int kBankTimeout = 1 << 29;    // a merge gave up waiting for a partial (also counted in the time-out word, ovf[1])
__device__ inline bool split_scaled(float x, _Float16& hi, _Float16& lo) {
  const float y = x * kBankScale;
  split_f16(y, hi, lo);
  return !(fabsf(y) <= 65504.0f);        // (also true for NaN)
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

namespace {
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
}
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
  b.vpart = reinterpret_cast<float*>(p);
  p += align256((size_t)no * Tcap * (b.hwp / kJT) * kDo * sizeof(float));
  b.colsum = reinterpret_cast<float*>(p);
  p += align256((size_t)no * Tcap * kDo * sizeof(float));
  b.area = reinterpret_cast<int32_t*>(p);
  p += align256((size_t)no * Tcap * 4);
  b.ovf = reinterpret_cast<int32_t*>(p);            // control block: overflow word, then the arrival tickets
  b.cnt = reinterpret_cast<int32_t*>(p + 256);
  return b;
}

size_t bank_ctl_bytes(int no, int h, int w) { return 256 + align256((size_t)no * bank_nqt_max(h * w) * 2 * 4); }

size_t bank_bytes(int no, int Tcap, int h, int w) {
  const size_t hwp = ((size_t)h * w + kJT - 1) / kJT * kJT;
  return 2 * (size_t)no * Tcap * hwp * kDe * 2 + 2 * (size_t)no * Tcap * kDo * hwp * 2 +
         align256((size_t)no * Tcap * (hwp / kJT) * kDo * 4) + align256((size_t)no * Tcap * kDo * 4) +
         align256((size_t)no * Tcap * 4) + bank_ctl_bytes(no, h, w);
}

size_t bank_area_offset(int no, int Tcap, int h, int w) {
  return bank_bytes(no, Tcap, h, w) - bank_ctl_bytes(no, h, w) - align256((size_t)no * Tcap * 4);
}

namespace {

// ------------------------------------------------------------------------------------------ append
// grid = (tiles, no * nf, 1 + kDo / kDe): blockIdx.y = o * nf + f writes slot slot0 + f of object o from
// frame f of the source: element (o, c, f, cell) of k4 at o * k_os + c * k_cs + f * hw + cell (same for
// v4).  The frame loop appends one frame of contiguous [no,C,h,w] tensors (nf = 1, k_cs = hw); the
// drop-in MemoryReader entry stages all T frames of a [no,C,T,h,w] memory in one launch.
__global__ __launch_bounds__(kThreads) void bk_append(BankView b, int slot0, const int32_t* __restrict__ slot_dev, int nf,
                                                      const float* __restrict__ k4,
                                                      const float* __restrict__ v4, long long k_cs,
                                                      long long k_os, long long v_cs, long long v_os,
                                                      const int32_t* __restrict__ rects) {
  __shared__ float tile[kJT][kDe + 1];
  const int u = blockIdx.x, tid = threadIdx.x;
  // slot_dev (optional): a device-resident frame counter added to slot0 -- lets a captured HIP graph of the frame
  // loop be replayed while the memory grows (the host never has to bake the slot into a kernel argument)
  if (slot_dev) slot0 += __builtin_amdgcn_readfirstlane(*slot_dev);
  if (slot0 < 0 || slot0 + nf > b.Tcap) {            // a full bank / a desynchronised device counter: nothing is written, and the
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) atomicOr(b.ovf, kBankBadSlot);   // bank says so
    return;                                          // (rmnet_bank_overflow_count() != 0: the caller must not trust its reads)
  }
  const int o = (int)blockIdx.y / nf, f = (int)blockIdx.y - o * nf, slot = slot0 + f;
  Rect rc{0, b.w - 1, 0, b.h - 1};
  if (rects) {
    const int32_t* r = rects + (size_t)blockIdx.y * 4;
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
  // grid.z = 1 + kDo / kDe: slice 0 converts the keys, slices 1.. one 128-channel chunk of the values
  // each (five short dependency chains instead of one long one: a launch has < 1 workgroup per CU)
  if (blockIdx.z == 0) {
    // keys: gather [c][cell] -> LDS [p][c] -> split -> [n][c]
    const float* kb = k4 + (size_t)o * k_os + (size_t)f * b.hw + cell;
#pragma unroll
    for (int i = 0; i < kDe / 8; ++i) {
      const int c = rg + 8 * i;
      tile[p][c] = valid ? kb[(size_t)c * k_cs] : 0.0f;
    }
    __syncthreads();
    {
      const int row = tid >> 3, c0 = (tid & 7) * 16;
      half8 h0, h1, l0, l1;
      bool over = false;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        _Float16 hi, lo;
        over |= split_scaled(tile[row][c0 + e], hi, lo);
        h0[e] = hi; l0[e] = lo;
        over |= split_scaled(tile[row][c0 + 8 + e], hi, lo);
        h1[e] = hi; l1[e] = lo;
      }
      if (over) atomicAdd(b.ovf, 1);
      const size_t off = ((so * b.hwp + (size_t)u * kJT + row) * kDe + c0) * sizeof(_Float16);
      *reinterpret_cast<half8*>(b.kh + off) = h0;
      *reinterpret_cast<half8*>(b.kh + off + 16) = h1;
      *reinterpret_cast<half8*>(b.kl + off) = l0;
      *reinterpret_cast<half8*>(b.kl + off + 16) = l1;
    }
    return;
  }
  // values: 128 channels at a time through the same LDS tile (coalesced gather along cells, like the
  // keys), then fragment order [tile u][d/16][lane = d%16 + 16 g][8 cells]: one thread builds one
  // whole 16-byte fragment row (channel d, lane group gg) -- its 8 cells are the compact offsets
  // {4gg..4gg+3, 16+4gg..16+4gg+3} (kperm) -- from 8 LDS reads and writes it with ONE
  // 16-byte store per plane; a wave covers 16 channels x 4 groups = 4 x 256 B runs.
  {
    const float* vb = v4 + (size_t)o * v_os + (size_t)f * b.hw + cell;
    const size_t vbase = (so * (b.hwp / kJT) + u) * (size_t)(kDo * kJT) * sizeof(_Float16);
    const int gg = tid & 3;
    {
      const int c0 = ((int)blockIdx.z - 1) * kDe;
#pragma unroll
      for (int i = 0; i < kDe / 8; ++i) {
        const int c = rg + 8 * i;
        tile[p][c] = valid ? vb[(size_t)(c0 + c) * v_cs] : 0.0f;
      }
      __syncthreads();
      if (tid < kDe) {   // this tile's column sums (cells in ascending order; padding cells are 0): the read-out of a
        float sum = 0.0f;   // masked query cell is the mean of m_val over ALL cells (bk_main's static part)
#pragma unroll 8
        for (int j = 0; j < kJT; ++j) sum += tile[j][tid];
        b.vpart[((so * (size_t)(b.hwp / kJT) + u) * kDo) + c0 + tid] = sum;
      }
#pragma unroll
      for (int i = 0; i < kDe / 64; ++i) {
        const int dl = (tid >> 2) + 64 * i, d = c0 + dl;
        half8 hi8, lo8;
        bool over = false;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = 4 * gg + (e & 3) + 16 * (e >> 2);
          _Float16 hi, lo;
          over |= split_scaled(tile[j][dl], hi, lo);
          hi8[e] = hi; lo8[e] = lo;
        }
        if (over) atomicAdd(b.ovf, 1);
        const size_t off = vbase + (size_t)(((d >> 4) * 64) + (d & 15) + 16 * gg) * 16;
        *reinterpret_cast<half8*>(b.vh + off) = hi8;
        *reinterpret_cast<half8*>(b.vl + off) = lo8;
      }
    }
  }
}

// grid = no * nf blocks of kDo threads: slot (o, slot0 + f)'s column sums = its tiles' sums in tile order.
__global__ __launch_bounds__(kDo) void bk_colsum(BankView b, int slot0, const int32_t* __restrict__ slot_dev, int nf) {
  if (slot_dev) slot0 += __builtin_amdgcn_readfirstlane(*slot_dev);
  if (slot0 < 0 || slot0 + nf > b.Tcap) return;      // (bk_append has flagged it)
  const int o = (int)blockIdx.x / nf, f = (int)blockIdx.x - o * nf, d = threadIdx.x;
  const size_t so = (size_t)o * b.Tcap + slot0 + f;
  const int ntiles = (b.area[so] + kJT - 1) / kJT;
  const float* __restrict__ src = b.vpart + so * (size_t)(b.hwp / kJT) * kDo + d;
  float sum = 0.0f;
  int u = 0;
  for (; u + 8 <= ntiles; u += 8) {       // eight loads in flight, added in tile order (a dependent load per tile cost 15 us)
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(u + k) * kDo];
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += v[k];
  }
  for (; u < ntiles; ++u) sum += src[(size_t)u * kDo];
  b.colsum[so * kDo + d] = sum;
}

// ------------------------------------------------------------------------------------------ read
struct BArgs {
  BankView b;
  const float *qk, *qv;
  const int32_t* qry_rects;  // [no][4] or null
  float* out;                // [no][2 * kDo][h][w]: read-out, then q_val x box (models/rmnet.py:163)
  float* ws_o;               // [no][slots] partial blocks in fragment order (common.h)
  float* ws_ml;              // [no][slots][2][kQT]: running reference (log2 domain) and sum
  int32_t* ws_plan;          // [no][kPlanInts]
  int T;
  const int32_t* T_dev;      // optional: device-resident frame counter added to T (graph replay while the memory grows)
  int gate;                  // != 0: do nothing when the bank's overflow word is set (mr_main then runs instead)
  int obj0, nobj;            // objects [obj0, obj0 + nobj) belong to this launch (nobj <= kMaxObj)
  int slot0, target;         // first partial slot of the launch; workgroups to aim for
  int Tmax;                  // frames readable through this view (<= kMaxT; the view may start at a later slot of a longer bank)
  float* ml_out;             // optional [no][2][h*w]: the soft-max state (reference m in the log2 domain, sum l) of every query cell
                             // that went through a merge -- lets the launcher chain reads of more than kMaxT frames (bk_chain)
  float qscale;              // log2(e) / sqrt(De) * 2^6, folded into the query fragments
};

constexpr int kKbuf = kJT * kDe * 2;                       // bytes of one K plane tile (8 KB)
constexpr int kPbuf = 4 * 2 * 64 * 16;                     // P fragments of one tile [ntile][hi/lo][lane] x 16 B
constexpr int kLdsBytes = 8 * kKbuf                        // K hi/lo x 4 ring slots
                          + 3 * kPbuf                      // P buffers (two in the split mode, three in the fp16 mode)
                          + 3 * kQT * 4                    // alpha [buf][16 queries][4 query tiles]
                          + (kMaxT + 4) * 4                // tile prefix
                          + kMaxT * 4                      // cells per frame
                          + 12 * 256;                      // landing patches of the L2 prefetch, one per wave (fp16 mode)
static_assert(kLdsBytes + 4096 <= kLdsBytesPerCU, "bk_main's LDS (tile rings + the prefix arrays of kMaxT frames + plan scratch) must fit one CU");
constexpr int kProducers = 4;                              // waves 0-3
constexpr int kConsumers = 8;                              // waves 4-11
constexpr int kCDT = kDo / 16 / kConsumers;                // d-tiles (16 value channels) per consumer: 4
constexpr int kRThreads = 64 * (kProducers + kConsumers);  // 768 = 12 waves = 3 per SIMD
constexpr int kProducerPrio = 2;   // static priority of the producer waves (s_setprio)
constexpr int kPfStepsPP = 10;   // fp16 modes: L2 prefetch distance in steps of two tiles (L2Prefetch)

// Workgroup = 12 waves (3 per SIMD), 64 compacted queries x one split of the tile list.
//   waves 0-3  ("producers", static priority): S = K^T Q for 16 queries each (24 MFMAs per 32-cell
//               tile), online soft-max in registers, P -> fp16 hi/lo B-fragments -> LDS; the MFMAs
//               of tile n+2 are interleaved with the soft-max VALU chain of tile n+1.
//   waves 4-11 ("consumers"): O += V P for 64 value channels x 64 queries each (48 MFMAs per tile),
//               V A-fragments straight from the bank into registers, each channel tile refilled
//               with the next tile right after its MFMAs; they also move the K tiles
//               bank -> registers -> swizzled LDS ring (they have the slack).
//   Wave i, i+4 and i+8 share a SIMD: the matrix pipe of every SIMD sees 24 + 2 x 48 = 120 MFMAs per
//   tile.  The producer's serial chain (LDS fragment reads -> MFMAs -> soft-max VALU -> LDS publish)
//   is shorter than that and runs one tile AHEAD of the PV, so the consumers never wait for it.
//   (The previous 8-wave version gave the producers a PV share as well; its trace showed the
//   producer chain = PV share + S + soft-max as the critical path of every tile, 2x the MFMA time.)
//   ONE barrier per tile.
struct Walk {          // per-workgroup constants of the tile walk (all wave-uniform)
  int jt0, ntl, qt, Mq;
  int o;               // object (absolute index)
  int slot;            // partial slot (absolute index)
  Rect qr;
  int t;               // frame of the first tile
  int pf_part, pf_nparts;   // fp16 mode: this workgroup's share of the L2 prefetch (0 parts: none)
};

// Wave-uniform cursor over the split's tile list.  seek(j) clamps j to the split's last tile, which
// makes every prefetch UNCONDITIONAL (a load past the end re-reads the last tile): with branches
// around the loads hipcc cannot count them and falls back to s_waitcnt vmcnt(0), which drains the
// loads issued a moment ago.  The frame boundary lives in registers; the LDS prefix is only read
// when a frame ends.
struct Cursor {
  int tt, base, nextb, jlast;
  const int* tpre;
  __device__ inline void init(const int* tp, int frame, int jl) {
    tpre = tp; tt = frame; jlast = jl;
    base = __builtin_amdgcn_readfirstlane(tp[frame]);
    nextb = __builtin_amdgcn_readfirstlane(tp[frame + 1]);
  }
  __device__ inline int seek(int j) {   // returns the tile's index inside its frame
    j = min(j, jlast);
    while (j >= nextb) {                // (also skips frames with an empty box)
      ++tt;
      base = nextb;
      nextb = __builtin_amdgcn_readfirstlane(tpre[tt + 1]);
    }
    return j - base;
  }
};


// A q_key element of a query INSIDE the box whose scaled value leaves fp16's window (|x| * qscale >= 65504, i.e. |q_key| beyond
// ~8e3), or a NaN / Inf, cannot be represented in the query fragments: it is counted in the bank's overflow word, like an
// out-of-window memory element, and the caller re-reads exactly (rmnet_hip.h).  The test runs on the fp16 fragments themselves
// (magnitude bits >= 65504: the split mode saturates there, the fp16 mode converts to Inf; NaN patterns are larger still) and
// AFTER the tile walk: in the producers' prologue the same ~100 instructions cost the workgroup 2.5 us of its critical path
// (r04, bench loop), after the walk they are free.
__device__ inline void query_range_check(const BankView& b, const half8 (&qh)[4], bool qvalid) {
  unsigned um = 0u;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const u32x4 d = __builtin_bit_cast(u32x4, qh[ks]);
#pragma unroll
    for (int i = 0; i < 4; ++i) um = max(um, max(d[i] & 0x7fffu, (d[i] >> 16) & 0x7fffu));
  }
  if (qvalid && um >= 0x7bffu) atomicAdd(b.ovf, 1);
}

// ---------------------------------------------------------------- producers: S and soft-max
// Barrier protocol (identical count in consumer_loop): A, B, then one per tile.
//   pre A        consumers: K(0..3) -> ring slots 0..3
//   between A, B producers: S(0), S(1), soft-max(0) -> P[0], fragments of K(2); consumers: request K(4)
//   iteration n  producers: S(n+2) INTERLEAVED with soft-max(n+1) -> P[(n+1)&1]; then the fragments
//                           of K(n+3) from slot (n+3)%4, so the next iteration starts with MFMAs
//                consumers: PV(n) from P[n&1]; K(n+4) -> slot n%4 (last read in iteration n-3)
// The S MFMAs of tile n+2 and the soft-max VALU chain of tile n+1 are independent, so inside the
// producer wave the matrix pipe and the VALU overlap instead of running one after the other.
__device__ inline void producer_loop(const BArgs& a, const Walk& wk, char* Kl_, char* Pl_, float* Al,
                                     const int* tpre, const int* tarea, int wave, int lane,
                                     float& m_out, float& l_out) {
  const BankView& b = a.b;
  const int o = wk.o;
  const int l15 = lane & 15, g = lane >> 4;
  const int jt0 = wk.jt0, ntl = wk.ntl;
  // query fragments (B operand of S): lane (query l15, group g) holds channels 32ks + 8g + e
  half8 qh[4], ql[4];
  {
    const int qn = wk.qt * kQT + wave * 16 + l15;
    const bool qvalid = qn < wk.Mq;
    int cell = 0;
    if (qvalid) {
      const int rw = wk.qr.width(), ry = qn / rw;
      cell = (wk.qr.cy0 + ry) * b.w + wk.qr.cx0 + (qn - ry * rw);
    }
    const float* qb = a.qk + (size_t)o * kDe * b.hw + cell;
    // 1/sqrt(De) (models/rmnet.py:156) and log2(e) are folded into the query: S comes out of the
    // MFMAs in the log2 domain (times 2^12, see kSraw) and the soft-max is one fma + v_exp_f32 (= 2^x)
    // per element.
    const float keep = qvalid ? a.qscale : 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = qb[(size_t)(32 * ks + 8 * g + e) * b.hw] * keep;
        _Float16 hi, lo;
        split_f16(x, hi, lo);              // (saturates; an element outside the window is COUNTED by a consumer wave: query_range_check)
        qh[ks][e] = hi; ql[ks][e] = lo;
      }
  }
  float mref = -INFINITY, lsum = 0.0f;

  // K ring feeding (was the consumers' job: its ~35 instructions sat on THEIR critical path -- a
  // consumer's head issues slowly while the other waves' MFMAs own the SIMD's issue slots -- whereas a
  // producer idles ~800 cycles per tile at the barrier).  A tile is one contiguous 8 KB block per plane
  // = 512 chunks of 16 B; producer thread pt moves chunks pt and pt + 256 of both planes.  LDS image:
  // row = cell (256 B), chunk c of row r stored at chunk c ^ (r & 15) -> the ds_read_b128 of the A
  // fragments are conflict-free.
  const size_t so0 = (size_t)o * b.Tcap;
  const int pt = wave * 64 + lane;
  const int prow = pt >> 4;
  const int kdst = prow * 256 + (((pt & 15) ^ (prow & 15)) << 4);      // (row prow + 16: + 4096, same swizzle)
  auto k_load = [&](half8 (&kr)[4], int tt, int ll) {
    const size_t off = ((so0 + tt) * b.hwp + (size_t)ll * kJT) * kDe * sizeof(_Float16) + (size_t)pt * 16;
    kr[0] = *reinterpret_cast<const half8*>(b.kh + off);
    kr[1] = *reinterpret_cast<const half8*>(b.kh + off + 4096);
    kr[2] = *reinterpret_cast<const half8*>(b.kl + off);
    kr[3] = *reinterpret_cast<const half8*>(b.kl + off + 4096);
  };
  auto k_store = [&](const half8 (&kr)[4], int slot) {
    char* d = Kl_ + slot * 2 * kKbuf + kdst;
    *reinterpret_cast<half8*>(d) = kr[0];
    *reinterpret_cast<half8*>(d + 4096) = kr[1];
    *reinterpret_cast<half8*>(d + kKbuf) = kr[2];
    *reinterpret_cast<half8*>(d + kKbuf + 4096) = kr[3];
  };
  half8 kr[4];                     // K tile n+5 on its way to the ring (one iteration to land)
  Cursor ck;
  ck.init(tpre, wk.t, jt0 + ntl - 1);
  {
    half8 k0[4], k1[4], k2[4], k3[4];
    const int a0 = ck.seek(jt0);
    k_load(k0, ck.tt, a0);
    const int a1 = ck.seek(jt0 + 1);
    k_load(k1, ck.tt, a1);
    const int a2 = ck.seek(jt0 + 2);
    k_load(k2, ck.tt, a2);
    const int a3 = ck.seek(jt0 + 3);
    k_load(k3, ck.tt, a3);
    const int a4 = ck.seek(jt0 + 4);
    k_load(kr, ck.tt, a4);
    k_store(k0, 0);
    k_store(k1, 1);
    k_store(k2, 2);
    k_store(k3, 3);
  }

  struct Frags { half8 a0h[4], a1h[4], a0l[4], a1l[4]; };
  auto k_frags = [&](Frags& f, int kslot) {   // 16 conflict-free ds_read_b128 (XOR-swizzled rows)
    const char* kb = Kl_ + kslot * 2 * kKbuf;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int sw = ((4 * ks + g) ^ l15) << 4;
      f.a0h[ks] = *reinterpret_cast<const half8*>(kb + l15 * 256 + sw);
      f.a1h[ks] = *reinterpret_cast<const half8*>(kb + (16 + l15) * 256 + sw);
      f.a0l[ks] = *reinterpret_cast<const half8*>(kb + kKbuf + l15 * 256 + sw);
      f.a1l[ks] = *reinterpret_cast<const half8*>(kb + kKbuf + (16 + l15) * 256 + sw);
    }
  };
  // S = K^T Q of one tile (log2 domain).  Two accumulator chains (cells 0-15 / 16-31), the three
  // split terms summed inside the chain, small terms first; consecutive MFMAs alternate chains, which
  // is all the distance a dependent 16x16x32 MFMA needs.  Lane result: S[cell 4g + r (+16)][query l15].
  auto s_mfma = [&](const Frags& f, f32x4& s0, f32x4& s1) {
    s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a0l[0], qh[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a1l[0], qh[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks > 0) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a0l[ks], qh[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a1l[ks], qh[ks], s1, 0, 0, 0);
      }
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a0h[ks], ql[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a1h[ks], ql[ks], s1, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a0h[ks], qh[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a1h[ks], qh[ks], s1, 0, 0, 0);
    }
  };
  // Online soft-max of one tile and its P fragments (k index of the fragment: e = r (+4)).
  // While the matrix pipe is saturated by the consumers a VALU instruction of this wave issues only
  // every ~11 cycles (trace), so the chain is kept short: no scale/log2e multiplies (folded into q),
  // the padding mask only on a frame's last tile, packed hi conversion, lo = p - hi in one
  // v_fma_mix per element.  nvalid = cells of this tile that exist (0 for the phantom tile past the
  // split's end: every score is -inf then, P = 0 and the running state is untouched).
  auto soft_max = [&](const f32x4& s0, const f32x4& s1, int nvalid, int pbuf) {
    float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    if (nvalid < kJT) {   // wave-uniform
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sv[r] = (4 * g + r < nvalid) ? sv[r] : -INFINITY;
        sv[4 + r] = (16 + 4 * g + r < nvalid) ? sv[4 + r] : -INFINITY;
      }
    }
    float tmax = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])),
                       fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    tmax = group4_max(tmax);
    // deferred running reference (raw units; first tile: mref = -inf): bump only when exceeded by > kDefer
    const bool bump = tmax > mref + kDeferRaw;
    const float alpha = bump ? __builtin_amdgcn_exp2f((mref - tmax) * kSraw) : 1.0f;
    mref = bump ? tmax : mref;
    const float nm = -mref * kSraw;
    float pv[8];
    float rs = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[e], kSraw, nm));   // in [0, e^8]: no saturation needed for the split
      rs += pv[e];
    }
    rs = group4_sum(rs);
    lsum = lsum * alpha + rs;
    u32x4 ph, plo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const half2 h = {(_Float16)pv[2 * i], (_Float16)pv[2 * i + 1]};
      const unsigned hk = __builtin_bit_cast(unsigned, h);
      unsigned lk;   // lo = fp16(p - hi): fp32 subtract and one rounding, straight into its half
      // hipcc does not pad hazards around inline asm (cdna_hip_programming.md 5.7 item 2).  The inputs come
      // straight from v_exp_f32 / v_cvt_pk (a transcendental result needs one wait state before a VALU reads
      // it) and v_fma_mixhi reads the register v_fma_mixlo has just written with a destination half-select
      // (one more): without the s_nops one producer wave occasionally published a wrong lo plane -- errors of
      // ~5e-3 on 16 queries of one launch in thirty (tests/stress_race.py).
      asm("s_nop 0\n\t"
          "v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
          "s_nop 0\n\t"
          "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
          "s_nop 0"
          : "=&v"(lk) : "v"(pv[2 * i]), "v"(pv[2 * i + 1]), "v"(hk));
      ph[i] = hk;
      plo[i] = lk;
    }
    char* pb = Pl_ + pbuf * kPbuf + wave * 2048 + lane * 16;
    *reinterpret_cast<u32x4*>(pb) = ph;
    *reinterpret_cast<u32x4*>(pb + 1024) = plo;
    if (g == 0) Al[pbuf * kQT + l15 * 4 + wave] = alpha;
  };

  Cursor cs;                       // tile whose soft-max comes next
  cs.init(tpre, wk.t, jt0 + ntl - 1);
  f32x4 sp0, sp1;                  // S of the tile after it (computed one iteration earlier)
  Frags f;
  __syncthreads();                                   // A: K tiles 0..3 in LDS
  {
    f32x4 s0, s1;
    k_frags(f, 0);
    s_mfma(f, s0, s1);
    k_frags(f, 1);
    s_mfma(f, sp0, sp1);
    const int l0 = cs.seek(jt0);
    soft_max(s0, s1, tarea[cs.tt] - l0 * kJT, 0);
    k_frags(f, 2);                                   // for the MFMAs of iteration 0
  }
  __syncthreads();                                   // B: P(0) visible
  int kslot = 3;                                     // ring slot of tile n+3
  for (int n = 0; n < ntl; ++n) {
    {
      const int l1 = cs.seek(jt0 + n + 1);
      const int nvalid = n + 1 < ntl ? tarea[cs.tt] - l1 * kJT : 0;
      f32x4 s0, s1;
      s_mfma(f, s0, s1);                             // tile n+2 (fragments read before the barrier)
      __builtin_amdgcn_sched_barrier(0);             // MFMAs first (the pipe is idle right after a barrier)

      soft_max(sp0, sp1, nvalid, (n + 1) & 1);       // tile n+1
      sp0 = s0; sp1 = s1;
      k_frags(f, kslot);                             // tile n+3: its LDS latency hides under the barrier
      {
        // K ring: tile n+4 (requested one iteration ago) -> slot n%4, whose last reader (tile n) passed
        // the barrier of iteration n-3; request tile n+5 (a clamped duplicate past the end: harmless)
        k_store(kr, (kslot + 1) & 3);
        const int l5 = ck.seek(jt0 + n + 5);
        k_load(kr, ck.tt, l5);
      }
    }
    kslot = (kslot + 1) & 3;
    __syncthreads();
  }
  m_out = mref * kSraw;                                // log2 domain; the segment's epilogue takes it from here
  l_out = lsum;
  query_range_check(b, qh, wk.qt * kQT + wave * 16 + l15 < wk.Mq);
}

// ---------------------------------------------------------------- consumers: O += V P, K ring
__device__ inline void consumer_loop(const BArgs& a, const Walk& wk, char* Kl_, char* Pl_, float* Al,
                                     const int* tpre, int wave, int lane,
                                     f32x4 (&acc)[kCDT][4]) {
  const BankView& b = a.b;
  const int o = wk.o;
  const int l15 = lane & 15, g = lane >> 4;
  const int jt0 = wk.jt0, ntl = wk.ntl;
  const size_t so0 = (size_t)o * b.Tcap;
  const size_t tiles_per_slot = (size_t)(b.hwp / kJT);
  // V: fragment-ordered planes; this wave's first d-tile and its lane inside a tile's 32 KB plane
  const int dt0 = kCDT * (wave - kProducers);
  const size_t vlane = (size_t)(dt0 * 64 + lane) * 16;
  auto v_tile = [&](int tt, int ll) { return ((so0 + tt) * tiles_per_slot + ll) * (size_t)(kDo * kJT * 2) + vlane; };
  // V fragment registers: ONE set.  The PV loop is channel-tile-major, so the fragments of channel
  // tile dt are dead after its 12 MFMAs and are refilled with the NEXT tile right there: the 8 + 2
  // loads of a tile are spread over the whole PV phase (issued together after the PV, the 80 loads
  // of the 8 consumers queue up in the CU's one address unit, 16 cycles each, while the matrix pipe
  // idles: that serial tail was 1/3 of a tile) and every load has one full iteration to land.
  half8 vh[kCDT], vl[kCDT];
  Cursor cv;
  cv.init(tpre, wk.t, jt0 + ntl - 1);
  {
    const int l0 = cv.seek(jt0);
    const size_t off = v_tile(cv.tt, l0);
#pragma unroll
    for (int dt = 0; dt < kCDT; ++dt) {
      vh[dt] = *reinterpret_cast<const half8*>(b.vh + off + dt * 1024);
      vl[dt] = *reinterpret_cast<const half8*>(b.vl + off + dt * 1024);
    }
  }
#pragma unroll
  for (int dt = 0; dt < kCDT; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[dt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                   // A: K tiles 0..3 visible
  __syncthreads();                                   // B: P(0) visible

  for (int n = 0; n < ntl; ++n) {
    const int buf = n & 1;
    const char* pfr = Pl_ + buf * kPbuf;
    const f32x4 al = *reinterpret_cast<const f32x4*>(Al + buf * kQT + l15 * 4);
    half8 bh[4], bl[4];                              // all P fragments of this tile
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      bh[it] = *reinterpret_cast<const half8*>(pfr + it * 2048 + lane * 16);
      bl[it] = *reinterpret_cast<const half8*>(pfr + it * 2048 + 1024 + lane * 16);
    }
    const int l1 = cv.seek(jt0 + n + 1);             // refill source: tile n+1 (clamped past the end)
    const size_t noff = v_tile(cv.tt, l1);
    const char* nvh = b.vh + noff;
    const char* nvl = b.vl + noff;
    if (__any(al[0] != 1.0f || al[1] != 1.0f || al[2] != 1.0f || al[3] != 1.0f)) {
      // (single v_mul_f32 on purpose, as in consumer_loop_pp: hipcc's packed v_pk_mul_f32 form of this rescale gave wrong read-outs
      //  in the ping-pong walk's timing -- profiles/r06_a_pingpong_walk.md)
#pragma unroll
      for (int dt = 0; dt < kCDT; ++dt)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[dt][it][r]) : "v"(al[it]));
    }
    // ---- O += V P: 4 channel tiles x 4 query tiles x 3 split terms; the 4 query tiles between two
    //      uses of an accumulator keep the MFMAs independent
#pragma unroll
    for (int dt = 0; dt < kCDT; ++dt) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[dt], bh[it], acc[dt][it], 0, 0, 0);
#pragma unroll
      for (int it = 0; it < 4; ++it)
        acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dt], bl[it], acc[dt][it], 0, 0, 0);
#pragma unroll
      for (int it = 0; it < 4; ++it)
        acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dt], bh[it], acc[dt][it], 0, 0, 0);
      {   // this channel tile's fragments of the next tile, unconditionally
        vh[dt] = *reinterpret_cast<const half8*>(nvh + dt * 1024);
        vl[dt] = *reinterpret_cast<const half8*>(nvl + dt * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE (the scheduler sinks them to the end)
    }
    __syncthreads();   // the one barrier per tile
  }

}

// ---- L2 prefetch of the fp16-operand walk.  Inside the frame loop nothing of the bank is in a cache when the read starts (the
// convolutions between two reads stream hundreds of MB), a V fragment load is issued ONE step (~1.3 us) before its use and an
// HBM miss under load takes longer than that: the tile walk then runs at memory latency, not at the matrix pipe's pace (measured
// in the loop: 1.77 us per step against 1.30 with a warm cache).  The nqt workgroups of a column block walk the same K / V tiles
// in lockstep on one XCD, i.e. behind one L2: each of them touches 1/nqt of the 128-byte lines of the step kPfStepsPP steps ahead
// (one dword per line) so that the demand loads of all of them hit the L2.  The touch is an LDS-DMA load into a 256-byte junk
// patch of the issuing wave: no destination register, nothing ever waits for it (hipcc does not see the load: its own counted
// waits only get more conservative; __syncthreads() stays a bare barrier).  In the loop the producers issue it (one wave
// instruction per step for the usual 12 query tiles); the first kPfStepsPP - 1 steps are touched by the CONSUMERS while they wait for
// the producers' first soft-max -- in the producers' own prologue those address computations sat on the critical path of the
// whole workgroup.
struct L2Prefetch {
  static constexpr int kLinesK = kJT * kDe * 2 / 128, kLinesV = kDo * kJT * 2 / 128, kLinesT = kLinesK + kLinesV;   // 64 + 256 per tile
  Cursor c;
  const char *kh, *vh;
  size_t so0, tps, hwp;
  int jt0, ntl, part, nparts;
  unsigned patch;
  __device__ inline void init(const BankView& b, const Walk& wk, const int* tpre, char* lds_base, int wave_slot) {
    c.init(tpre, wk.t, wk.jt0 + wk.ntl - 1);
    kh = b.kh; vh = b.vh;
    so0 = (size_t)wk.o * b.Tcap; tps = (size_t)(b.hwp / kJT); hwp = (size_t)b.hwp;
    jt0 = wk.jt0; ntl = wk.ntl; part = wk.pf_part; nparts = wk.pf_nparts;
    patch = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds_base + kLdsBytes - 12 * 256 + wave_slot * 256));
  }
  // the lines of step `step`, this workgroup's share, spread over waves w = 0 .. nw - 1
  __device__ inline void touch(int step, int w, int nw, int lane) {
    if (nparts <= 0) return;
    const int ja = jt0 + 2 * step;
    if (ja >= jt0 + ntl) return;                                         // (wave-uniform)
    const int la = c.seek(ja);
    const char* ka = kh + ((so0 + c.tt) * hwp + (size_t)la * kJT) * kDe * sizeof(_Float16);
    const char* va = vh + ((so0 + c.tt) * tps + la) * (size_t)(kDo * kJT * 2);
    const int lb = c.seek(ja + 1);                                       // (clamped to the segment's last tile)
    const char* kb2 = kh + ((so0 + c.tt) * hwp + (size_t)lb * kJT) * kDe * sizeof(_Float16);
    const char* vb2 = vh + ((so0 + c.tt) * tps + lb) * (size_t)(kDo * kJT * 2);
    const int nl = (2 * kLinesT + nparts - 1) / nparts;                  // lines of this workgroup
    for (int l0 = w * 64; l0 < nl; l0 += nw * 64) {                      // (wave-uniform trip count)
      int i = (l0 + lane) * nparts + part;
      i = i < 2 * kLinesT ? i : part;                                    // (past the share: a duplicate of its first line)
      const bool second = i >= kLinesT;
      const int r = second ? i - kLinesT : i;
      const bool isk = r < kLinesK;
      const char* base = second ? (isk ? kb2 : vb2) : (isk ? ka : va);
      const char* ad = base + (size_t)(isk ? r : r - kLinesK) * 128;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(ad), "s"(patch) : "memory");
    }
  }
};

// ================================================================================================================
// [r6] PING-PONG walk of the fp16-operand modes.  The lockstep walk above synchronises its 12 waves once per step and
// every wave is in the same phase at the same time: both consumers of a SIMD issue their MFMAs together, then read
// their P fragments together, then wait -- matrix pipe, L1 and LDS take turns instead of overlapping (3,350 cycles
// per step for 1,312 of MFMA, 1,280 of L1 and ~600 of LDS; r05_g counters: pipe busy 28-32 %, waiting 46 %).
// Here a step is TWO barrier intervals and the two consumers of a SIMD work half a step apart:
//     interval E(n): group A (waves 4-7, channels 0-255)   O += V(n) P(n), 32 MFMAs, V(n+1) requested behind each channel tile
//                    group B (waves 8-11, channels 256-511) "load phase": P(n) LDS -> registers, its share of K(n+6) -> ring,
//                                                           K(n+7) requested, addresses of V(n+1)
//     interval O(n): B   O += V(n) P(n);   A   load phase: P(n+1), K(n+6), K(n+7), addresses of V(n+2)
// so the matrix pipe of a SIMD always has one consumer's 32 MFMAs (+ the producer's 8-10) while the other consumer's
// LDS reads, ring stores and address arithmetic run in their shadow.  The producer's step is cut at the tile: per
// interval it issues the 8 S MFMAs of ONE tile, re-reads that tile's fragment registers with the tile of the next
// step (the LDS round trip flies under the rest of the interval instead of ending it) and does half of a soft-max:
//     E(m): soft-max part 2 of step m+1 (exp2, round, publish P(m+1), denominator) | S of tile B of step m+2 | fragments A(m+3)
//     O(m): S of tile A of step m+3 | fragments B(m+3) | soft-max part 1 of step m+2 (mask, row max, reference, alpha)
// (ONE fragment register set: a k-step's two registers are re-read with the next tile right behind their last MFMA.)
// Hand-offs (all by the one barrier between two intervals): P(s) is written in E(s-1), read by A in O(s-1) and by B in
// E(s) (buffer s % 3); K(s) is stored in E/O(s-6), its tile A read in E(s-3), its tile B in O(s-3) (ring slot s & 3).
// Barrier count per segment: A, B, 2 nst + 1 -- identical in all three roles (B runs A's loop one interval late).
// ================================================================================================================
template <bool kQx>
__device__ inline void producer_loop_pp(const BArgs& a, const Walk& wk, char* Kl_, char* Pl_, float* Al,
                                        const int* tpre, const int* tarea, int wave, int lane,
                                        float& m_out, float& l_out) {
  const BankView& b = a.b;
  const int o = wk.o;
  const int l15 = lane & 15, g = lane >> 4;
  const int jt0 = wk.jt0, ntl = wk.ntl;
  const int nst = (ntl + 1) >> 1;
  half8 qh[4], ql[kQx ? 4 : 1];
  {
    const int qn = wk.qt * kQT + wave * 16 + l15;
    const bool qvalid = qn < wk.Mq;
    int cell = 0;
    if (qvalid) {
      const int rw = wk.qr.width(), ry = qn / rw;
      cell = (wk.qr.cy0 + ry) * b.w + wk.qr.cx0 + (qn - ry * rw);
    }
    const float* qb = a.qk + (size_t)o * kDe * b.hw + cell;
    const float keep = qvalid ? a.qscale : 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = qb[(size_t)(32 * ks + 8 * g + e) * b.hw] * keep;   // (range: query_range_check)
        if constexpr (kQx) {
          _Float16 hi, lo;
          split_f16(x, hi, lo);
          qh[ks][e] = hi; ql[ks][e] = lo;
        } else {
          qh[ks][e] = (_Float16)x;
        }
      }
  }
  float mref = -INFINITY, lsum = 0.0f;
  half8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;

  // The L2 touches of the walk stay with the PRODUCERS (measured in a consumer's load phase, r6: an LDS-DMA load that misses to HBM
  // sits in that wave's in-order vmcnt queue in front of its next V fragments and stalls the following PV phase by its whole
  // latency; a producer never waits for a vector memory operation).
  L2Prefetch pf;
  pf.init(b, wk, tpre, Kl_, wave);

  struct Half { half8 r0[4], r1[4]; };                  // one tile: cells 0-15 / 16-31, four k-steps
  auto k_half = [&](Half& f, int kslot, int plane) {    // 8 conflict-free ds_read_b128 (XOR-swizzled rows)
    const char* kb = Kl_ + kslot * 2 * kKbuf + plane * kKbuf;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int sw = ((4 * ks + g) ^ l15) << 4;
      f.r0[ks] = *reinterpret_cast<const half8*>(kb + l15 * 256 + sw);
      f.r1[ks] = *reinterpret_cast<const half8*>(kb + (16 + l15) * 256 + sw);
    }
  };
  struct S2 { f32x4 s[2]; };                            // S of one tile: cells 4g + r / 16 + 4g + r of query l15
  // S of one tile: two accumulator chains, interleaved (qx: the query's lo plane first at every k-step).  With kReload the two
  // fragment registers of a k-step are re-read from (kslot, plane) right behind their last MFMA: the LDS round trip of the
  // NEXT step's tile starts in the first half of the interval and never ends it.
  auto s_half = [&](Half& f, S2& r, bool reload, int kslot, int plane) {
    const char* kb = Kl_ + kslot * 2 * kKbuf + plane * kKbuf;
    r.s[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    r.s[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (kQx) {
        r.s[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.r0[ks], ql[ks], r.s[0], 0, 0, 0);
        r.s[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.r1[ks], ql[ks], r.s[1], 0, 0, 0);
      }
      r.s[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.r0[ks], qh[ks], r.s[0], 0, 0, 0);
      r.s[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.r1[ks], qh[ks], r.s[1], 0, 0, 0);
      if (reload) {
        const int sw = ((4 * ks + g) ^ l15) << 4;
        f.r0[ks] = *reinterpret_cast<const half8*>(kb + l15 * 256 + sw);
        f.r1[ks] = *reinterpret_cast<const half8*>(kb + (16 + l15) * 256 + sw);
      }
    }
  };
  // scheduling pattern of an interval's region: per k-step {MFMA, VALU x kV} x (2 or 4), then the two fragment reads
  auto interleave = [&]() {
    constexpr int kV = kQx ? 3 : 6;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int i = 0; i < (kQx ? 4 : 2); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kV, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
  };
  // soft-max, part 1: padding mask, row maximum, deferred reference, rescale factor.  Leaves nm = -reference (log2 domain)
  // and alpha for part 2; the masked scores stay in sa / sb.
  float nm = 0.0f, alpha = 1.0f;
  auto soft_max1 = [&](S2& sa, S2& sb, int nva, int nvb) {
    if (nva < kJT || nvb < kJT) {                       // (wave-uniform: its own basic block)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sa.s[0][i] = (4 * g + i < nva) ? sa.s[0][i] : -INFINITY;
        sa.s[1][i] = (16 + 4 * g + i < nva) ? sa.s[1][i] : -INFINITY;
        sb.s[0][i] = (4 * g + i < nvb) ? sb.s[0][i] : -INFINITY;
        sb.s[1][i] = (16 + 4 * g + i < nvb) ? sb.s[1][i] : -INFINITY;
      }
    }
    float tmax = fmaxf(fmaxf(fmaxf(sa.s[0][0], sa.s[0][1]), fmaxf(sa.s[0][2], sa.s[0][3])),
                       fmaxf(fmaxf(sa.s[1][0], sa.s[1][1]), fmaxf(sa.s[1][2], sa.s[1][3])));
    tmax = fmaxf(tmax, fmaxf(fmaxf(fmaxf(sb.s[0][0], sb.s[0][1]), fmaxf(sb.s[0][2], sb.s[0][3])),
                             fmaxf(fmaxf(sb.s[1][0], sb.s[1][1]), fmaxf(sb.s[1][2], sb.s[1][3]))));
    tmax = group4_max(tmax);
    const bool bump = tmax > mref + kDeferRaw;
    alpha = bump ? __builtin_amdgcn_exp2f((mref - tmax) * kSraw) : 1.0f;
    mref = bump ? tmax : mref;
    nm = -mref * kSraw;
  };
  // soft-max, part 2: weights, rounding to fp16 (B-fragment layout of O = V P), publication, denominator (the sum of the
  // ROUNDED weights: an MFMA against an all-ones fragment)
  auto soft_max2 = [&](const S2& sa, const S2& sb, int pbuf) {
    u32x4 pa, pb_;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const half2 h0 = {(_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sa.s[0][2 * i], kSraw, nm)),
                        (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sa.s[0][2 * i + 1], kSraw, nm))};
      const half2 h1 = {(_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sa.s[1][2 * i], kSraw, nm)),
                        (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sa.s[1][2 * i + 1], kSraw, nm))};
      const half2 h2 = {(_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sb.s[0][2 * i], kSraw, nm)),
                        (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sb.s[0][2 * i + 1], kSraw, nm))};
      const half2 h3 = {(_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sb.s[1][2 * i], kSraw, nm)),
                        (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sb.s[1][2 * i + 1], kSraw, nm))};
      pa[i] = __builtin_bit_cast(unsigned, h0);
      pa[2 + i] = __builtin_bit_cast(unsigned, h1);
      pb_[i] = __builtin_bit_cast(unsigned, h2);
      pb_[2 + i] = __builtin_bit_cast(unsigned, h3);
    }
    char* pb = Pl_ + pbuf * kPbuf + wave * 2048 + lane * 16;
    *reinterpret_cast<u32x4*>(pb) = pa;
    *reinterpret_cast<u32x4*>(pb + 1024) = pb_;
    if (g == 0) Al[pbuf * kQT + l15 * 4 + wave] = alpha;
    f32x4 lc = {lsum * alpha, 0.f, 0.f, 0.f};
    lc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, __builtin_bit_cast(half8, pa), lc, 0, 0, 0);
    lc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, __builtin_bit_cast(half8, pb_), lc, 0, 0, 0);
    lsum = lc[0];
  };
  // cell counts of a step (for the padding mask of a frame's last tile).  The cursor keeps the CURRENT frame's area in an SGPR: the
  // LDS is read only when a frame ends, so the per-step look-up is scalar arithmetic (as a per-step LDS -> SGPR round trip it sat
  // in the producers' O interval, the longest of the walk: r6 interval time line)
  Cursor cs;
  cs.init(tpre, wk.t, jt0 + ntl - 1);
  int cs_area = __builtin_amdgcn_readfirstlane(tarea[wk.t]), cs_tt = wk.t;
  auto valid_of = [&](int j) {                       // cells of tile j that exist (0 past the split's end)
    const int l = cs.seek(j);
    if (cs.tt != cs_tt) { cs_tt = cs.tt; cs_area = __builtin_amdgcn_readfirstlane(tarea[cs_tt]); }   // (wave-uniform; once per frame)
    return j < jt0 + ntl ? cs_area - l * kJT : 0;
  };
  auto step_valid = [&](int step, int& nva, int& nvb) {
    nva = valid_of(jt0 + 2 * step);
    nvb = valid_of(jt0 + 2 * step + 1);
  };
  auto bar = [&]() {               // (LDS only crosses an interval boundary; the L2 touches are never waited for)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  int nva0, nvb0, nva1, nvb1, nva, nvb;
  step_valid(0, nva0, nvb0);       // (LDS -> SGPR round trips taken while the consumers' first K tiles are on their way)
  step_valid(1, nva1, nvb1);
  step_valid(2, nva, nvb);
  Half f;                          // ONE fragment set: it alternates between tile B of a step (read in O, used in E) and tile A of
                                   // the next step (read in E, used in O) -- every register is re-read right behind its last MFMA
  S2 ca, cb;                       // S of the step whose soft-max part 2 comes next (part 1 done)
  S2 na, nb;                       // S of the step after it (tile A complete; tile B computed in E)
  __syncthreads();                                   // A: K steps 0..3 in the ring
  {
    S2 s0a, s0b;
    k_half(f, 0, 0);
    s_half(f, s0a, true, 0, 1);                      // S(0) tile A; tile B of step 0 behind it
    s_half(f, s0b, true, 1, 0);                      //      tile B; tile A of step 1
    s_half(f, ca, true, 1, 1);                       // S(1)
    s_half(f, cb, true, 2, 0);
    soft_max1(s0a, s0b, nva0, nvb0);
    soft_max2(s0a, s0b, 0);                          // P(0)
    s_half(f, na, true, 2, 1);                       // S of tile A of step 2; tile B of step 2 behind it: E(0)
    soft_max1(ca, cb, nva1, nvb1);                   // part 1 of step 1 (part 2: E(0))
  }
  __syncthreads();                                   // B: P(0) visible; ring slots 0, 1 free (K steps 4, 5 go there in E(0))
  int pbuf = 1;                                      // (m + 1) % 3
  // One step = E(m), O(m).  The three S register sets rotate (cur <- next <- new): the loop body is written three times with the
  // names rotated instead of copying 24 registers per step.
  S2 ta, tb;                       // third set (tile A of step m+3 lands in ta)
  auto step_pair = [&](int m, S2& c_a, S2& c_b, S2& n_a, S2& n_b, S2& t_a) {
    // ---- E(m)
    __builtin_amdgcn_sched_barrier(0);
    s_half(f, n_b, true, (m + 3) & 3, 0);            // S of tile B of step m+2; tile A of step m+3 behind it
    soft_max2(c_a, c_b, pbuf);                       // step m+1 -> P(m+1)
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    pbuf = pbuf == 2 ? 0 : pbuf + 1;
    bar();
    // ---- O(m)
    s_half(f, t_a, true, (m + 3) & 3, 1);            // S of tile A of step m+3; tile B of step m+3 behind it
    soft_max1(n_a, n_b, nva, nvb);                   // step m+2
    interleave();
    __builtin_amdgcn_sched_barrier(0);
    step_valid(m + 3, nva, nvb);
    pf.touch(m + kPfStepsPP, wave, kProducers, lane);
    bar();
  };
  for (int m = 0; m < nst; m += 3) {
    step_pair(m, ca, cb, na, nb, ta);
    if (m + 1 >= nst) break;
    step_pair(m + 1, na, nb, ta, tb, ca);
    if (m + 2 >= nst) break;
    step_pair(m + 2, ta, tb, ca, cb, na);
  }
  bar();                                             // (the consumers' last interval)
  m_out = mref * kSraw;
  l_out = lsum;
  query_range_check(b, qh, wk.qt * kQT + wave * 16 + l15 < wk.Mq);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the prefetch patch is free again)
}

__device__ inline void consumer_loop_pp(const BArgs& a, const Walk& wk, char* Kl_, char* Pl_, float* Al,
                                        const int* tpre, int wave, int lane,
                                        f32x4 (&acc)[kCDT][4]) {
  const BankView& b = a.b;
  const int o = wk.o;
  const int l15 = lane & 15;
  const int jt0 = wk.jt0, ntl = wk.ntl;
  const int nst = (ntl + 1) >> 1;
  const bool late = wave >= kProducers + kConsumers / 2;   // group B: one interval behind group A (wave-uniform)
  const size_t so0 = (size_t)o * b.Tcap;
  const size_t tiles_per_slot = (size_t)(b.hwp / kJT);
  const int dt0 = kCDT * (wave - kProducers);
  const size_t vlane = (size_t)(dt0 * 64 + lane) * 16;
  auto v_tile = [&](int tt, int ll) { return ((so0 + tt) * tiles_per_slot + ll) * (size_t)(kDo * kJT * 2) + vlane; };
  // K ring as in the lockstep walk: slot = step & 3, plane = tile of the step; consumer thread ct moves chunk ct of both tiles
  const int ct = (wave - kProducers) * 64 + lane;
  const int crow = ct >> 4;
  const int kdst = crow * 256 + (((ct & 15) ^ (crow & 15)) << 4);
  Cursor ck;
  ck.init(tpre, wk.t, jt0 + ntl - 1);
  auto k_load = [&](half8 (&kr)[2], int step) {      // both tiles of a step (clamped past the end)
    const int la = ck.seek(jt0 + 2 * step);
    kr[0] = *reinterpret_cast<const half8*>(b.kh + ((so0 + ck.tt) * b.hwp + (size_t)la * kJT) * kDe * sizeof(_Float16) + (size_t)ct * 16);
    const int lb = ck.seek(jt0 + 2 * step + 1);
    kr[1] = *reinterpret_cast<const half8*>(b.kh + ((so0 + ck.tt) * b.hwp + (size_t)lb * kJT) * kDe * sizeof(_Float16) + (size_t)ct * 16);
  };
  auto k_store = [&](const half8 (&kr)[2], int slot) {
    char* d = Kl_ + slot * 2 * kKbuf + kdst;
    *reinterpret_cast<half8*>(d) = kr[0];
    *reinterpret_cast<half8*>(d + kKbuf) = kr[1];
  };
  half8 k4[2], k5[2], kr[2];       // K steps 4, 5 (stored behind barrier B) and the step on its way to the ring
  {
    half8 k0[2], k1[2], k2[2], k3[2];
    k_load(k0, 0);
    k_load(k1, 1);
    k_load(k2, 2);
    k_load(k3, 3);
    k_load(k4, 4);
    k_load(k5, 5);
    k_store(k0, 0);
    k_store(k1, 1);
    k_store(k2, 2);
    k_store(k3, 3);
  }
  half8 va[kCDT], vb[kCDT];        // V fragments of the step's two tiles
  Cursor cv;
  cv.init(tpre, wk.t, jt0 + ntl - 1);
  {
    const int la = cv.seek(jt0);
    const size_t offa = v_tile(cv.tt, la);
    const int lb = cv.seek(jt0 + 1);
    const size_t offb = v_tile(cv.tt, lb);
#pragma unroll
    for (int dt = 0; dt < kCDT; ++dt) {
      va[dt] = *reinterpret_cast<const half8*>(b.vh + offa + dt * 1024);
      vb[dt] = *reinterpret_cast<const half8*>(b.vh + offb + dt * 1024);
    }
  }
#pragma unroll
  for (int dt = 0; dt < kCDT; ++dt)
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[dt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 pa[4], pb[4];              // P fragments of the step this wave multiplies next
  f32x4 al;                        // its rescale factors
  auto p_frags = [&](int buf) {
    const char* pfr = Pl_ + buf * kPbuf;
    al = *reinterpret_cast<const f32x4*>(Al + buf * kQT + l15 * 4);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      pa[it] = *reinterpret_cast<const half8*>(pfr + it * 2048 + lane * 16);
      pb[it] = *reinterpret_cast<const half8*>(pfr + it * 2048 + 1024 + lane * 16);
    }
  };
  auto bar = [&]() {               // LDS crosses an interval boundary, the V / K requests stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  const char *nva, *nvb;           // addresses of the NEXT step's V tiles (found in the load phase)
  auto v_next = [&](int step) {
    const int la = cv.seek(jt0 + 2 * step);          // (clamped past the end)
    nva = b.vh + v_tile(cv.tt, la);
    const int lb = cv.seek(jt0 + 2 * step + 1);
    nvb = b.vh + v_tile(cv.tt, lb);
  };
  int kstep = 6;                   // K step stored in this wave's next load phase (ring slot kstep & 3); kstep + 1 is requested there
  int pb3 = 0;                     // pstep % 3 of the next load phase
  auto load_phase = [&](int pstep) {                 // pstep: the step this wave multiplies next
    p_frags(pb3);
    pb3 = pb3 == 2 ? 0 : pb3 + 1;
    k_store(kr, kstep & 3);
    ++kstep;
    k_load(kr, kstep);
    v_next(pstep + 1);
  };
  __syncthreads();                                   // A: K steps 0..3 in the ring
  k_load(kr, 6);
  {                                                  // (idle until B: the producers compute S(0..2), P(0))
    L2Prefetch pf;
    pf.init(b, wk, tpre, Kl_, wave);
#pragma unroll 1
    for (int s_ = 1; s_ < kPfStepsPP; ++s_) pf.touch(s_, wave - kProducers, kConsumers, lane);
  }
  __syncthreads();                                   // B: P(0) visible; ring slots 0, 1 free
  k_store(k4, 0);                                    // visible before O(0), where the producers read tile A of step 4
  k_store(k5, 1);
  if (late) {                                        // group B: E(0) is a load phase
    load_phase(0);
    bar();
  } else {
    p_frags(0);
    pb3 = 1;
    v_next(1);
  }
  for (int n = 0; n < nst; ++n) {
    // ---- O += V(n) P(n)
    if (__any(al[0] != 1.0f || al[1] != 1.0f || al[2] != 1.0f || al[3] != 1.0f)) {
      // (single v_mul_f32 on purpose: as `acc[dt][it] *= al[it]` hipcc packs the 64 multiplies into v_pk_mul_f32 with op_sel, and in
      //  this walk the read-out of 16 queries x one channel row then came out wrong in ~9 of 10 reads of tests/stress_race.py's 8-object
      //  case -- always the low halves of the `it = 1` pairs of one accumulator, lanes 48-63; with plain multiplies 0 of 8.
      //  profiles/r06_a_pingpong_walk.md has the bisect.  The packed form is also the slower one beside MFMAs: MI355X_MICROARCH.md)
#pragma unroll
      for (int dt = 0; dt < kCDT; ++dt)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[dt][it][r]) : "v"(al[it]));
    }
#pragma unroll
    for (int dt = 0; dt < kCDT; ++dt) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va[dt], pa[it], acc[dt][it], 0, 0, 0);
#pragma unroll
      for (int it = 0; it < 4; ++it)
        acc[dt][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[dt], pb[it], acc[dt][it], 0, 0, 0);
      va[dt] = *reinterpret_cast<const half8*>(nva + dt * 1024);
      vb[dt] = *reinterpret_cast<const half8*>(nvb + dt * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    bar();
    // ---- load phase for step n+1 (under the other group's MFMAs)
    load_phase(n + 1);
    bar();
  }
  if (!late) bar();                                  // (group B's last load phase)
}

constexpr int kMaxObj = kBankMaxObj;    // objects planned together in one launch (the launcher groups more)

// Partial slots of one (object, query tile) pair: a strided run (the aligned column blocks) followed by a run of
// consecutive slots (the remainder chunks that touch the pair) -- see bank_chunks() in common.h.
struct PairSlots {
  int a0, na, sa, b0, count, cf;
  __device__ inline int slot(int s_) const { return s_ < na ? a0 + s_ * sa : b0 + (s_ - na); }
};
__device__ inline PairSlots pair_slots(int sb, int nqt, const BankChunks& bc, int qt) {
  PairSlots r{sb + qt, bc.nfull, nqt, 0, bc.nfull, 0};
  if (bc.R > 0) {
    const int v0 = qt * (bc.R + bc.sc);                                   // the pair on the virtual line
    r.cf = plan_div(v0, bc.C);
    const int cl = plan_div(v0 + bc.R - 1, bc.C);                            // remainder chunks touching it
    r.b0 = sb + nqt * bc.nfull + r.cf + qt;
    r.count += cl - r.cf + 1;
  }
  return r;
}

__device__ inline float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- static part of the read: the outputs that do not depend on the soft-max --------------------------------------
//   out[o][kDo + d][cell] = q_val[o][d][cell] x box(o)        (cat(mem, q_val), models/rmnet.py:163; the box of :358)
//   out[o][d][cell]       = mean_j m_val[o][d][j]            for the query cells OUTSIDE the box: all their logits are 0,
//                           the soft-max is uniform over all T*h*w memory cells (file header of memory_read.hip);
//                           = sum of the slots' column sums (bk_append / bk_colsum) / (T*h*w).  Also written for the
//                           cells inside the box when nothing is memorised inside any memory box (no pair, no merge).
// It is ~2.6x the q_val bytes of HBM traffic and no arithmetic: measured as a prologue of every workgroup it cost
// 11-14 us (a bandwidth-bound burst in front of the tile loops); as a WORK QUEUE (one ticket word in the bank) it is
// drained by a few workgroups that get no chunk (the plan sets them aside, scaled to the launch) while the others
// compute, and by every workgroup that has finished its segments -- among them the early arrivers of a pair, while
// the pair's last arriver merges.
constexpr int kStaticRowsPerTicket = 2;            // (object, channel) rows a wave takes per ticket
constexpr float kStaticBytesPerUs = 35.0e3f;             // what one streaming workgroup moves (sizing of the set-aside)
constexpr float kStaticBytesPerUsF16 = 50.0e3f;
constexpr float kTileUs = 1.75f, kTileUsF16 = 0.55f, kLaunchUs = 12.0f;   // tile step / fixed part of a compute workgroup (same estimate)
static_assert(kSplitMax * kQT * 4 <= 2 * kPbuf, "merge weights live in the P buffers");
static_assert(kConsumers * 16 * 65 * 4 + 2 * 4 * kQT * 4 <= 8 * kKbuf, "epilogue scratch lives in the K ring");

// =========================================== [r6] merge of a pair by ALL its workgroups ===========================================
// A pair cut into nsp segments has nsp published partials (O as 128 fragment UNITS of 16 channels x 16 queries = 1 KB each, m, l).
// Until round 5 the pair's last arriver merged them alone: nsp - 1 dependent trips to memory of 128 KB each, 2.2-2.6 us per trip
// (profiles/r04_f_dropin_breakdown.md: 20-23 us of a 51 us dense read while 230 workgroups had nothing to do).  Now the workgroup
// at position s of the pair owns the units U = s, s + n, s + 2 n, ... (n = the pair's aligned column blocks -- the workgroups that
// run ONE segment -- or all its segments if it has none; spread over the workgroup's eight consumer waves), waits until all nsp
// partials are published, and reads its units from EVERY slot in one batch of independent loads, issued before the weights exist:
// the merge is one trip to memory whatever nsp is.  The sum runs over the slots in slot order -- the read-out does not depend on who arrived when.
// The wait needs the pair's other workgroups to be running or finished; they never wait inside their walks, the launch has at most
// one workgroup per CU (kSplitTargetSlots <= 256 CUs), and the poll is bounded (time-out word + NaN read-out, never a hang).
struct MergeJob {
  int o, qt, sself;          // object (absolute), query tile, this workgroup's position among the pair's segments
  int nqt, njt, C, own;      // the object's plan: query tiles, memory tiles, chunk length, kind of plan (bank_chunks)
  int slot_obj;              // the object's first partial slot
  Rect qr;                   // query box
  float n_out;               // masked memory cells of the object
  // (the launch's pointers BY VALUE: a reference to the kernel's argument block would move the whole block to the stack)
  float *ws_o, *ws_ml, *out, *ml_out;
  int32_t *cnt, *ovf;
  int hw, w;
};
// (NOT inlined: inside bk_main its 24 loads in flight pushed hipcc into spilling inside the tile walks)
template <int kTerms>
__device__ __forceinline__ void merge_pair(const MergeJob m, char* Kl_, char* Pl_, int* sgave) {
  const int tid = threadIdx.x, hw = m.hw;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < kProducers;
  const int ln = tid & 63, l15 = ln & 15, g = ln >> 4;
  const BankChunks bc = bank_chunks(m.nqt, m.njt, m.C, seg_cost_of(kTerms), m.own);
  const PairSlots ps = pair_slots(m.slot_obj, m.nqt, bc, m.qt);
  const int nsp = ps.count, qt = m.qt, sself = m.sself;
  const int nown = ps.na > 0 ? ps.na : nsp;               // the pair's merging workgroups: its aligned column blocks if it has any (a
                                                          // remainder chunk runs several segments; it only publishes), else all
  const int o = m.o;
  const Rect qr = m.qr;
  const int Mq = qr.area();
  const float n_out = m.n_out;                            // masked memory cells: S = 0, V = 0 (file header)
  auto sld = [](const int& x) { return __builtin_amdgcn_readfirstlane(x); };   // LDS value -> SGPR
  constexpr size_t kSlotF = (size_t)kDo * kQT;            // floats per partial slot
  auto slot_rsrc = [&](int slot) {                        // buffer descriptor of a partial slot (wave-uniform base)
    float* base = m.ws_o + (size_t)slot * kSlotF;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(base));
    const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(base) >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uintptr_t)bhi << 32) | blo), 0,
                                             (int)(kSlotF * 4), 0x00020000);
  };
  float* Wt = reinterpret_cast<float*>(Pl_);              // [slot of the pair][query]: 2^(m_s - m_tot)   (16 KB: the P buffers)
  float* red = reinterpret_cast<float*>(Kl_);             // [4][64] partial maxima                       (the idle K ring)
  float* red2 = red + 4 * kQT;                            // [4][64] partial sums
  int* left = m.cnt + 2 * ((size_t)o * bank_nqt_max(hw) + qt);
  int* done = left + 1;
  if (tid == 0) {
    int polls = 0;
    bool gave_up = false;
    while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nsp) {
      __builtin_amdgcn_s_sleep(2);
      if (++polls > (1 << 21)) { gave_up = true; break; }
    }
    *sgave = gave_up ? 1 : 0;
    if (gave_up) {   // (a workgroup of the pair never ran.  Never hang the GPU: count it in the time-out word, make the bank say
      atomicAdd(m.ovf + 1, 1);                                          // "do not trust me" and leave the counters alone; the
      atomicOr(m.ovf, kBankTimeout);                                    // launcher clears the control block before every read)
    } else if (__hip_atomic_fetch_add(left, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nown - 1) {
      __hip_atomic_store(left, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the last one to see the pair complete: clean
      __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // counters for the next read
    }
  }
  __syncthreads();                                                                // M1: all partials of the pair are in memory
  const float poison = sld(*sgave) ? __builtin_nanf("") : kBankValueUnscale;
  //      Producers: the pair's weights.  thread (sl = wave, qi = lane): query qi, slots sl, sl + 4, ...
  //      The partials were stored write-through and are read with sc1 loads (L2-coherent at agent scope: no acquire fence, no L1
  //      invalidate needed).
  constexpr int kMl = 4;
  float m_r[kMl], l_r[kMl];
  auto ml_load = [&](int sj, int qi, float& m_, float& l_) {
    const float* e = m.ws_ml + (size_t)ps.slot(sj) * 2 * kQT;
    m_ = __hip_atomic_load(e + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    l_ = __hip_atomic_load(e + kQT + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (producer) {
    const int qi = ln, sl = wave;
    float mloc = -INFINITY;
#pragma unroll
    for (int j = 0; j < kMl; ++j) {
      const int sj = sl + 4 * j;
      m_r[j] = -INFINITY; l_r[j] = 0.0f;
      if (sj < nsp) ml_load(sj, qi, m_r[j], l_r[j]);
      mloc = fmaxf(mloc, m_r[j]);
    }
    for (int sj = sl + 4 * kMl; sj < nsp; sj += 4) {
      float ms, ls;
      ml_load(sj, qi, ms, ls);
      mloc = fmaxf(mloc, ms);
    }
    red[sl * kQT + qi] = mloc;
    __syncthreads();                                                              // M2
    float mtot = fmaxf(fmaxf(red[qi], red[kQT + qi]), fmaxf(red[2 * kQT + qi], red[3 * kQT + qi]));
    if (n_out > 0.0f) mtot = fmaxf(mtot, 0.0f);          // the masked memory cells have S = 0
    float lloc = 0.0f;
#pragma unroll
    for (int j = 0; j < kMl; ++j) {
      const int sj = sl + 4 * j;
      const float wgt = __builtin_amdgcn_exp2f(m_r[j] - mtot);     // (-inf for a slot that does not exist: 0)
      if (sj < nsp) Wt[sj * kQT + qi] = wgt;
      lloc += l_r[j] * wgt;
    }
    for (int sj = sl + 4 * kMl; sj < nsp; sj += 4) {
      float ms, ls;
      ml_load(sj, qi, ms, ls);
      const float wgt = __builtin_amdgcn_exp2f(ms - mtot);
      Wt[sj * kQT + qi] = wgt;
      lloc += ls * wgt;
    }
    // (the closed-form term N_out * 2^(-m_tot) of the masked memory cells rides in group 0's partial sum)
    red2[sl * kQT + qi] = lloc + (sl == 0 && n_out > 0.0f ? n_out * __builtin_amdgcn_exp2f(-mtot) : 0.0f);
    __syncthreads();                                                              // M3
    if (m.ml_out && wave == 0 && sself == 0) {   // chained reads (> kMaxT frames): the state of this pair's queries
      const int nq = qt * kQT + ln;
      if (nq < Mq) {
        const int rw = qr.width(), ry = nq / rw;
        const int cell = (qr.cy0 + ry) * m.w + qr.cx0 + (nq - ry * rw);
        float* ml = m.ml_out + (size_t)o * 2 * hw;
        ml[cell] = mtot;
        ml[hw + cell] = red2[ln] + red2[kQT + ln] + red2[2 * kQT + ln] + red2[3 * kQT + ln];
      }
    }
    return;
  }
  // ---- consumers: this workgroup's units.  Unit U = 4 * (channel tile) + (query group); wave wv takes the workgroup's units
  //      number wv, wv + 8, ...  UN units x SN slots are in flight at a time (static register arrays; the shape follows nsp).
  //      The first batch of loads is issued BEFORE the weights exist (barriers M2 / M3 are passed behind it).
  const int wv = wave - kProducers;
  const int n_s = plan_div(128 - sself + nown - 1, nown);   // units of this workgroup
  constexpr int kOOB = 0x40000000;                       // a byte offset no slot reaches: the buffer unit returns 0 / drops the access
  bool synced = false;
  auto weights_ready = [&]() {
    if (!synced) { __syncthreads(); __syncthreads(); synced = true; }             // M2, M3 (consumer side)
  };
  auto merge_units = [&](auto un_tag, auto sn_tag) {
    constexpr int UN = decltype(un_tag)::value, SN = decltype(sn_tag)::value;
    for (int i0 = 0; wv + 8 * i0 < n_s; i0 += UN) {
      f32x4 r[UN];
      int U[UN];
#pragma unroll
      for (int ui = 0; ui < UN; ++ui) {
        const int j = wv + 8 * (i0 + ui);
        U[ui] = j < n_s ? sself + j * nown : -1;
        r[ui] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      for (int s0 = 0; s0 < nsp; s0 += SN) {
        u32x4 v[UN][SN];
#pragma unroll
        for (int sj = 0; sj < SN; ++sj) {
          const bool sv = s0 + sj < nsp;
          const __amdgpu_buffer_rsrc_t rs = slot_rsrc(ps.slot(sv ? s0 + sj : nsp - 1));
#pragma unroll
          for (int ui = 0; ui < UN; ++ui)
            v[ui][sj] = __builtin_amdgcn_raw_buffer_load_b128(rs, (sv && U[ui] >= 0) ? U[ui] * 1024 + ln * 16 : kOOB, 0, 16 /* sc1 */);
        }
        weights_ready();
#pragma unroll
        for (int sj = 0; sj < SN; ++sj) {
          const int sw = min(s0 + sj, nsp - 1) * kQT + l15;          // (a slot past the pair: its loads returned 0)
#pragma unroll
          for (int ui = 0; ui < UN; ++ui) r[ui] += Wt[sw + (U[ui] & 3) * 16] * __builtin_bit_cast(f32x4, v[ui][sj]);
        }
      }
#pragma unroll
      for (int ui = 0; ui < UN; ++ui) {
        if (U[ui] < 0) continue;                                       // (wave-uniform)
        const int q = (U[ui] & 3) * 16 + l15, nq = qt * kQT + q;
        const float iq = poison / (red2[q] + red2[kQT + q] + red2[2 * kQT + q] + red2[3 * kQT + q]);
        if (nq < Mq) {
          const int rw = qr.width(), ry = nq / rw;
          const int cell = (qr.cy0 + ry) * m.w + qr.cx0 + (nq - ry * rw);
          // (a store instruction covers 16 consecutive compacted queries of four channel rows)
          float* __restrict__ outo = m.out + ((size_t)o * 2 * kDo + 4 * (U[ui] >> 2) * 4 + 4 * g) * hw + cell;
#pragma unroll
          for (int e = 0; e < 4; ++e) outo[(size_t)e * hw] = r[ui][e] * iq;
        }
      }
    }
  };
  using std::integral_constant;
  if (nsp == 2) merge_units(integral_constant<int, 8>{}, integral_constant<int, 2>{});
  else if (nsp == 3) merge_units(integral_constant<int, 6>{}, integral_constant<int, 3>{});
  else if (nsp == 4) merge_units(integral_constant<int, 4>{}, integral_constant<int, 4>{});
  else if (nsp <= 6) merge_units(integral_constant<int, 4>{}, integral_constant<int, 6>{});
  else if (nsp <= 8) merge_units(integral_constant<int, 3>{}, integral_constant<int, 8>{});
  else if (nsp <= 12) merge_units(integral_constant<int, 2>{}, integral_constant<int, 12>{});
  else merge_units(integral_constant<int, 1>{}, integral_constant<int, 20>{});
  weights_ready();                                         // (a wave without units)
}

// Launch-wide work list (stream-K with L2-friendly order).  Per object the work is the matrix
// nqt(o) query tiles x njt(o) memory tiles.  ONE chunk length C (tiles per workgroup) is chosen for
// the whole launch so that the chunks of all objects fill the compute workgroups, whatever the box sizes.
// Object o is cut into
//   * nfull = njt / C column blocks of exactly C tiles: nqt * nfull chunks of ONE segment each,
//     (block, query tile) in block-major order -- the nqt workgroups of a block walk the same K/V
//     tiles in lockstep and sit on one XCD, so a tile comes from HBM once per L2 (measured: a plain
//     query-tile-major stream-K order, where neighbours read different tiles, is 1.4-1.9x slower);
//   * the remainder block of R = njt mod C tiles: its nqt * R tiles, query-tile-major, are cut into
//     chunks of C again; such a chunk crosses query tiles and runs several SEGMENTS (all inside the
//     same R <= C tile columns, which fit the L2).
// Every segment owns a partial slot.  The only segment of a pair writes the pair's read-out from its registers.  One of several
// publishes its partial (O, m, l) with write-through stores and counts itself done on the pair's counter; when a workgroup's
// walks are over it merges, for every pair it holds a segment of, ITS share of the pair's 128 fragment units from all the pair's
// partials (merge_pair below: [r6] until round 5 the pair's last arriver merged alone, one 128 KB trip to memory per partial).
template <int kTerms>
__global__ __launch_bounds__(kRThreads, 1) void bk_main(const BArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  __shared__ int o_njt[kMaxObj], o_m[kMaxObj], o_nqt[kMaxObj], o_cb[kMaxObj], o_sb[kMaxObj];
  __shared__ int o_rect[kMaxObj][4];
  __shared__ int plan_n, plan_c, plan_own, plan_g, sflag, sgave;
  __shared__ int o_c[kMaxObj];                      // chunk length of each object (one value for the launch unless it runs in rounds)
  char* Kl_ = lds;                                 // [ring slot][plane][8 KB]
  char* Pl_ = lds + 8 * kKbuf;                     // [buf][ntile][plane][lane*16]
  float* Al = reinterpret_cast<float*>(Pl_ + 3 * kPbuf);
  int* tpre = reinterpret_cast<int*>(Al + 3 * kQT);
  int* tarea = tpre + kMaxT + 4;

  const BankView& b = a.b;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < kProducers;
  if (a.gate && __builtin_amdgcn_readfirstlane(*b.ovf) != 0) return;   // out-of-window element: exact fp32 path runs
  if (tid == 0) sgave = 0;                                              // (set by a merge that timed out; read after several barriers)
  if (producer) __builtin_amdgcn_s_setprio(kProducerPrio);
  const int ng = a.nobj;
  const int hw = b.hw;
  // The frame count may live on the device (graph replay): its load, the areas of the first 64 slots and the query
  // rectangles are requested TOGETHER -- one memory round trip in front of the plan instead of two dependent ones
  // (inside the frame loop all three miss every cache).
  const int lane0 = tid & 63;
  int spec_ar = 0;
  if (a.nobj <= kProducers + kConsumers && wave < a.nobj && lane0 < a.Tmax)
    spec_ar = b.area[(size_t)(a.obj0 + wave) * b.Tcap + lane0];
  // [r6] the bank's sticky "largest affinity logit so far" word (below): requested here with the other early loads, compared after
  // the walk -- a stale value only costs a redundant atomic
  const int smax_seen = __hip_atomic_load(b.ovf + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int T_raw = a.T + (a.T_dev ? __builtin_amdgcn_readfirstlane(*a.T_dev) : 0);
  const int T_ = min(max(T_raw, 1), a.Tmax);         // memorised frames to read (the clamp is memory safety only:
  if (T_raw != T_ && blockIdx.x == 0 && tid == 0) atomicOr(b.ovf, kBankBadSlot);   // an out-of-range count is flagged)
  int* q_head = b.ovf + 16;                        // control block of the bank: static work queue (next item); zeroed by the launcher before every launch

  // ---- launch-wide plan, computed identically by every workgroup from the device-resident boxes
  //      (no host sync)
  // Fast path (<= 12 objects, <= 64 memorised frames): wave w owns object w, lane t its frame t; the
  // areas stay in registers, so the owner wave later builds the object's tile prefix without a
  // second trip to memory.  Otherwise: LDS atomics now, a reload of the object's areas later.
  const bool fastplan = ng <= kProducers + kConsumers && T_ <= RMNET_WAVE;
  int my_ar = 0;
  if (fastplan) {
    if (wave < ng && lane0 < T_) my_ar = spec_ar;
  } else if (tid < ng) {
    o_njt[tid] = 0; o_m[tid] = 0;
  }
  if (tid < ng) {   // query rectangle -> compacted queries -> query tiles
    int q0 = 0, q1 = b.w - 1, q2 = 0, q3 = b.h - 1;
    if (a.qry_rects) {
      const int32_t* q = a.qry_rects + (size_t)(a.obj0 + tid) * 4;
      q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
    }
    const Rect r{max(q0, 0), min(q1, b.w - 1), max(q2, 0), min(q3, b.h - 1)};
    const int Mq = r.area();
    o_rect[tid][0] = r.cx0; o_rect[tid][1] = r.cx1; o_rect[tid][2] = r.cy0; o_rect[tid][3] = r.cy1;
    o_nqt[tid] = (Mq + kQT - 1) / kQT;
  }
  if (fastplan) {
    const int tiles = wave_sum_fast((my_ar + kJT - 1) / kJT), cells = wave_sum_fast(my_ar);
    if (lane0 == 0 && wave < ng) { o_njt[wave] = tiles; o_m[wave] = cells; }
  } else {
    __syncthreads();
    for (int idx = tid; idx < ng * T_; idx += kRThreads) {
      const int og = idx / T_, t = idx - og * T_;
      const int ar = b.area[(size_t)(a.obj0 + og) * b.Tcap + t];
      atomicAdd(&o_njt[og], (ar + kJT - 1) / kJT);
      atomicAdd(&o_m[og], ar);
    }
  }
  __syncthreads();
  if (tid < RMNET_WAVE) {   // one wave, lane = object
    const int nqt = tid < ng ? o_nqt[tid] : 0, njt = tid < ng ? o_njt[tid] : 0;
    const int W = wave_sum_fast(nqt * njt), njt_max = wave_max_fast(njt);
    // workgroups set aside for the static part: what it takes to stream it in about the time the others compute
    // (the finishers drain whatever is left, so a wrong guess costs little either way)
    int target = a.target;
    {
      const float static_bytes = (float)ng * (float)hw * (float)kDo * 4.0f * 2.5f;
      const float compute_us = kLaunchUs + (kTerms != 3 ? kTileUsF16 : kTileUs) * (float)W / (float)a.target;
      int aside = (int)(static_bytes / (compute_us * (kTerms != 3 ? kStaticBytesPerUsF16 : kStaticBytesPerUs)) + 0.5f);
      aside = min(aside, a.target / 4);
      target = a.target - aside;
    }
    // [r6] The plan (common.h: bank_plan_pick).  64 candidates of each kind are priced AT ONCE, one per lane, in one loop over the
    // objects (the search sits on the critical path of every workgroup: until round 5 it walked the candidates four at a time with
    // wave reductions -- 5-6 rounds at the bench launch): lane i prices the i-th equalised candidate (common.h:
    // bank_eq_candidate) and -- if needed -- the plain plan at chunk length Clo + i * step (from the even cut to the longest object).
    constexpr int kSC = seg_cost_of(kTerms);
    constexpr int kCq = kTerms != 3 ? 2 : 1;              // (fp16 modes: a step is two tiles, an odd chunk wastes half of one)
    const int cmin = (bank_chunk_min(njt_max) + kCq - 1) / kCq * kCq;
    int Clo = max(W < (1 << 22) ? plan_div(W + target - 1, target) : (W + target - 1) / target, cmin);
    Clo = (Clo + kCq - 1) / kCq * kCq;
    const int ce = bank_eq_candidate(tid, njt_max, Clo, kCq, cmin);
    const int Chi = max((njt_max + kCq - 1) / kCq * kCq, Clo);
    const int step = (max(plan_div(Chi - Clo + 62, 63), 1) + kCq - 1) / kCq * kCq;
    const int cp = Clo + tid * step;
    const int nqe = njt > 0 ? nqt : 0;
    const int P = wave_sum_fast(nqe);
    int C0 = Clo, blocks = 0;
    if (P <= target) {
      int me = 0;
      for (int o = 0; o < ng; ++o)      // (lane o holds object o's tile counts: a readlane per object, no LDS round trip)
        me += bank_eq_count(__builtin_amdgcn_readlane(nqt, o), __builtin_amdgcn_readlane(njt, o), ce);
      const int c1n = wave_max_fast(me <= target ? -ce : -0x3fffffff);          // smallest chunk length whose equalised chunks fit
      const int c1 = c1n > -0x3fffffff ? -c1n : 0;
      const int cw = __builtin_amdgcn_readlane(ce, 0);                          // whole pairs (they fit: P <= target)
      // The plain plan's chunk length C0 is >= the even cut Clo: when the equalised plan wins against Clo it wins against C0 and the
      // plain candidates (three divisions per object and lane) are not priced at all -- the bench launch's case
      BankPlanPick pk = bank_plan_pick(c1, cw, Clo, kSC, kTerms);
      if (pk.blocks == 0) {
        int mp = 0;
        for (int o = 0; o < ng; ++o)
          mp += bank_chunks(__builtin_amdgcn_readlane(nqt, o), __builtin_amdgcn_readlane(njt, o), cp, kSC, 0).nch;
        const unsigned long long fp = __ballot(mp <= target);
        pk = bank_plan_pick(c1, cw, fp ? __builtin_amdgcn_readlane(cp, __builtin_ctzll(fp)) : 0, kSC, kTerms);
      }
      C0 = pk.C > 0 ? pk.C : Chi;
      blocks = pk.blocks;
      if (!blocks && wave_sum_fast(bank_chunks(nqt, njt, C0, kSC, 1).nch) <= target) blocks = 1;   // short objects as blocks of their own (common.h)
    }
    // [r6] more pairs than workgroups: ROUNDS of aligned chunks (common.h: bank_round_chunk_len) -- every object gets its own chunk
    // length, every chunk is one segment, a workgroup runs chunk c, c + G, c + 2 G, ...
    int Cobj = C0;
    if (P > target) {
      const int ps = wave_scan_incl_fast(nqe);
      const int Pw = wave_max_fast(ps <= plan_div(P, target) * target ? ps : 0);
      Cobj = bank_round_chunk_len(njt, ps, P, Pw, target, kCq);
      blocks = 2;
    }
    const BankChunks bc0 = bank_chunks(nqt, njt, Cobj, kSC, blocks);
    // chunks; partial slots (a remainder chunk can add one per query tile; an object that is ONE column block publishes nothing)
    const int my_ch = bc0.nch, my_sl = (P > target && bc0.nfull <= 1) ? 0 : bc0.nch + (bc0.R > 0 ? nqt : 0);
    const int nch = wave_scan_incl_fast(my_ch), nsl = wave_scan_incl_fast(my_sl);
    if (tid < ng) { o_cb[tid] = nch - my_ch; o_sb[tid] = nsl - my_sl; o_c[tid] = Cobj; }
    if (tid == RMNET_WAVE - 1) { plan_n = nch; plan_c = C0; plan_own = blocks; plan_g = target; }
  }
  __syncthreads();
  auto sld = [](const int& x) { return __builtin_amdgcn_readfirstlane(x); };   // LDS value -> SGPR
  const int nchunks = sld(plan_n), G = min(sld(plan_g), nchunks);   // chunks of the launch; workgroups that compute (the others: static part)
  if ((int)blockIdx.x < ng && tid == 0) {   // plan record of object blockIdx.x (tools / debugging only)
    const int og = blockIdx.x;
    int32_t* pr = a.ws_plan + (size_t)(a.obj0 + og) * kPlanInts;
    const Rect r{o_rect[og][0], o_rect[og][1], o_rect[og][2], o_rect[og][3]};
    pr[0] = r.area(); pr[1] = o_nqt[og]; pr[2] = o_njt[og]; pr[3] = o_m[og];
    pr[4] = r.cx0; pr[5] = r.cx1; pr[6] = r.cy0; pr[7] = r.cy1;
    pr[8] = a.slot0 + o_sb[og]; pr[9] = o_c[og]; pr[10] = nchunks; pr[11] = 1;
  }

  constexpr size_t kSlotF = (size_t)kDo * kQT;            // floats per partial slot
  auto slot_rsrc = [&](int slot) {                        // buffer descriptor of a partial slot (wave-uniform base)
    float* base = a.ws_o + (size_t)slot * kSlotF;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(base));
    const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(base) >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uintptr_t)bhi << 32) | blo), 0,
                                             (int)(kSlotF * 4), 0x00020000);
  };

  // [r6] merge pass over chunk c (after the walks): the same chunk -> segments enumeration as compute() below, one merge_pair per
  // segment whose pair has several
  auto merge_chunk = [&](const int c) {
    const int li = tid & 63;
    const unsigned long long le = __ballot((li < ng ? o_cb[li] : 0x7fffffff) <= c);
    const int og = le ? 63 - __builtin_clzll(le) : 0;
    MergeJob m;
    m.o = a.obj0 + og;
    m.nqt = sld(o_nqt[og]); m.njt = sld(o_njt[og]); m.C = sld(o_c[og]); m.own = sld(plan_own);
    m.slot_obj = a.slot0 + sld(o_sb[og]);
    m.qr = Rect{sld(o_rect[og][0]), sld(o_rect[og][1]), sld(o_rect[og][2]), sld(o_rect[og][3])};
    m.n_out = (float)(T_ * hw - sld(o_m[og]));
    m.ws_o = a.ws_o; m.ws_ml = a.ws_ml; m.out = a.out; m.ml_out = a.ml_out; m.cnt = b.cnt; m.ovf = b.ovf; m.hw = hw; m.w = b.w;
    const BankChunks bc = bank_chunks(m.nqt, m.njt, m.C, seg_cost_of(kTerms), m.own);
    const int cl = c - sld(o_cb[og]);
    if (cl < m.nqt * bc.nfull) {
      const int blk = plan_div(cl, m.nqt);
      m.qt = cl - blk * m.nqt; m.sself = blk;
      // (exactly one aligned block, or two segments in all: merged in place by the pair's last arriver, run_segment)
      if (bc.nfull >= 2 && pair_slots(m.slot_obj, m.nqt, bc, m.qt).count > 2) merge_pair<kTerms>(m, Kl_, Pl_, &sgave);
      return;
    }
    const int cr = cl - m.nqt * bc.nfull;
    const int u0 = cr * m.C, u1 = u0 + m.C, span = bc.R + bc.sc;
    if (bc.nfull > 0) return;                             // (pairs with aligned column blocks are merged by those)
    for (int qt = plan_div(u0, span); qt < m.nqt && qt * span < u1; ++qt) {
      if (min(u1 - qt * span, bc.R) <= max(u0 - qt * span, 0)) continue;
      m.qt = qt; m.sself = bc.nfull + cr - plan_div(qt * span, m.C);
      if (pair_slots(m.slot_obj, m.nqt, bc, qt).count > 2) merge_pair<kTerms>(m, Kl_, Pl_, &sgave);
    }
  };

  // =========================================== compute: this workgroup's chunk ===========================================
  auto compute = [&](const int c) {
  // object of chunk c = the last one whose chunk base is <= c (objects without chunks share the next one's base: the later one
  // wins); lane i looks at object i, and the nine plan values of the object travel LDS -> registers -> SGPRs in ONE round trip
  // (a serial search + a dozen dependent LDS reads cost ~1 us here)
  int li = tid & 63;
  asm volatile("" : "+v"(li));   // opaque per chunk: inside the rounds loop hipcc hoists the ten per-lane LDS addresses below out of the
                                 // loop and SPILLS them -- ten scratch round trips in front of every chunk's walk (r6: +5 us in the loop)
  const bool lv = li < ng;
  const int v_cb = lv ? o_cb[li] : 0x7fffffff, v_nqt = lv ? o_nqt[li] : 0, v_njt = lv ? o_njt[li] : 0, v_sb = lv ? o_sb[li] : 0;
  const int v_m = lv ? o_m[li] : 0, v_r0 = lv ? o_rect[li][0] : 0, v_r1 = lv ? o_rect[li][1] : 0, v_r2 = lv ? o_rect[li][2] : 0;
  const int v_r3 = lv ? o_rect[li][3] : 0, v_c = lv ? o_c[li] : 1;
  const unsigned long long le = __ballot(v_cb <= c);
  const int og = le ? 63 - __builtin_clzll(le) : 0;
  auto pick = [&](int v) { return __builtin_amdgcn_readlane(v, og); };
  const int nqt = pick(v_nqt), njt = pick(v_njt);
  const int C = pick(v_c);                  // this object's chunk length
  const BankChunks bc = bank_chunks(nqt, njt, C, seg_cost_of(kTerms), sld(plan_own));
  const int cl = c - pick(v_cb);            // chunk inside the object
  const int lane = tid & 63;
  Walk wk;
  wk.o = a.obj0 + og;
  wk.qr = Rect{pick(v_r0), pick(v_r1), pick(v_r2), pick(v_r3)};
  wk.Mq = wk.qr.area();
  const int slot_obj = a.slot0 + pick(v_sb);
  const float n_out = (float)(T_ * hw - pick(v_m));         // masked memory cells: S = 0, V = 0 (file header)

  // ---- this object's tile prefix over the T memorised frames
  if (fastplan) {
    if (wave == og) {   // the wave that holds this object's areas
      const int sc = wave_scan_incl_fast((my_ar + kJT - 1) / kJT);
      if (lane0 < T_) { tpre[lane0 + 1] = sc; tarea[lane0] = my_ar; }
      if (lane0 == 0) tpre[0] = 0;
    }
  } else if (tid < RMNET_WAVE) {
    int carry = 0;
    for (int base = 0; base < T_; base += RMNET_WAVE) {
      const int t = base + tid;
      const int ar = t < T_ ? b.area[(size_t)wk.o * b.Tcap + t] : 0;
      const int sc = wave_scan_incl_fast((ar + kJT - 1) / kJT);
      if (t < T_) { tpre[t + 1] = carry + sc; tarea[t] = ar; }
      carry += __builtin_amdgcn_readlane(sc, RMNET_WAVE - 1);
    }
    if (tid == 0) tpre[0] = 0;
  }
  __syncthreads();

  // One segment = the tile walk (producer / consumer roles) + its epilogue.  `sself` = this segment's position in
  // the pair's slot list.
  auto run_segment = [&](int sself) {
    int lo = 0, hi = T_;   // frame of the first tile: the last t with tpre[t] <= jt0 (tpre is non-decreasing, tpre[0] = 0)
    if (T_ <= RMNET_WAVE) {
      const int lt = tid & 63;
      lo = __popcll(__ballot(lt < T_ && tpre[lt] <= wk.jt0)) - 1;
    } else {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sld(tpre[mid]) <= wk.jt0) lo = mid; else hi = mid;
      }
    }
    wk.t = lo;
    int ln = lane;
    asm volatile("" : "+v"(ln));   // opaque per segment: keeps hipcc from hoisting the per-lane address
                                   // arithmetic of the loops below out of the segment loop (it then spills)
    f32x4 acc[kCDT][4];            // consumers: O of (64 channels x 64 queries)
    float m_seg = 0.0f, l_seg = 0.0f;   // producers: running reference (log2 domain) and sum of query 16 * wave + l15
    if constexpr (kTerms != 3) {          // 1: fp16 operands, 2: the same with an exact query -- one pipeline
      if (producer)
        producer_loop_pp<kTerms == 2>(a, wk, Kl_, Pl_, Al, tpre, tarea, wave, ln, m_seg, l_seg);
      else
        consumer_loop_pp(a, wk, Kl_, Pl_, Al, tpre, wave, ln, acc);
    } else {
      if (producer)
        producer_loop(a, wk, Kl_, Pl_, Al, tpre, tarea, wave, ln, m_seg, l_seg);
      else
        consumer_loop(a, wk, Kl_, Pl_, Al, tpre, wave, ln, acc);
    }

    // [r6] Largest soft-max reference of the segment's queries -> the bank's logit word (ovf[2], float bits, log2 domain, sticky
    // maximum; rmnet_hip.h).  The fp16-operand arithmetics are accurate while the logits are small (the rounding of K and q costs
    // ~2^-11 |S| per logit): the frame loop reads this word once per clip and re-reads the clip in the split arithmetic when it is
    // beyond the calibrated bound (rmnet_amd/rmnet.py 'auto'; profiles/r06_iou_temperature.md).  m_seg is the DEFERRED reference:
    // within 8 (natural units) below the true maximum.  Positive floats order like their bit patterns; in steady state the
    // maximum is already there and no atomic is issued.
    if (producer) {
      const int bits = wave_max_fast(__float_as_int(fmaxf(m_seg, 0.0f)));
      if (ln == 0 && bits > smax_seen) __hip_atomic_fetch_max(b.ovf + 2, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ================= segment epilogue (all 12 waves; every barrier below is reached by all of them) =================
    const int l15 = ln & 15, g = ln >> 4;
    const PairSlots ps = pair_slots(slot_obj, nqt, bc, wk.qt);
    const int nsp = ps.count;
    // epilogue scratch: merge weights in the P buffers (16 KB), the rest in the (now idle) K ring
    float* Wt = reinterpret_cast<float*>(Pl_);            // [slot of the pair][query]: 2^(m_s - m_tot)
    float* Tw = reinterpret_cast<float*>(Kl_);            // [consumer wave][16 channels][65]: transposes
    float* red = Tw + kConsumers * 16 * 65;               // [4][64] partial maxima
    float* red2 = red + 4 * kQT;                          // [4][64] partial sums
    float* Msh = Al;                                      // own (m, l) of the 64 queries
    float* Lsh = Al + kQT;
    const int dt0 = kCDT * (wave - kProducers);           // (consumers) first channel tile of this wave
    // [r6] Who merges a pair of several segments:
    //   * a pair with exactly ONE aligned column block (the plain plan: one block + remainder chunks), and a pair of TWO segments:
    //     its LAST ARRIVER (a ticket on the pair's first counter), HERE, with its own partial in registers -- the others hold a
    //     ticket, i.e. they have left their walks and only store: the wait for their publications cannot deadlock, whatever is
    //     resident -- and adds them slot by slot (two segments: one trip for the other's 128 KB; measured: 1.6 us less than
    //     publishing both and merging half each; a FIXED owner instead of the last arriver waits for the later one half of the
    //     time: +4 % at 8 / 12 object-frames);
    //   * every other pair (three or more equalised blocks, rounds, remainder-only pairs): all its workgroups (or its aligned ones)
    //     together, after their walks (merge_pair): this segment publishes and goes on.
    if (producer && g == 0) { Msh[wave * 16 + l15] = m_seg; Lsh[wave * 16 + l15] = l_seg; }
    bool owner_here = false;
    int* const pair_cnt = b.cnt + 2 * ((size_t)wk.o * bank_nqt_max(hw) + wk.qt);
    if (nsp > 1 && (ps.na == 1 || nsp == 2)) {
      if (tid == 0) sflag = __hip_atomic_fetch_add(pair_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();                                                              // E1: the ticket (and Msh / Lsh)
      owner_here = sld(sflag) == nsp - 1;
    }
    if (nsp > 1 && !owner_here) {
      // ---- publish with write-through (sc1) stores, drained by every storing wave, THEN count it
      //      (cdna_hip_programming.md section 6 Guideline 16, recipe R1 in its counter form)
      if (!producer) {
        const __amdgpu_buffer_rsrc_t rs = slot_rsrc(wk.slot);
#pragma unroll
        for (int dt = 0; dt < kCDT; ++dt)
#pragma unroll
          for (int it = 0; it < 4; ++it)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[dt][it]), rs,
                                                   (int)(partial_frag_offset(dt0 + dt, it, ln) * 4), 0, 16 /* sc1 */);
      } else if (g == 0) {
        float* wm = a.ws_ml + (size_t)wk.slot * 2 * kQT;
        __hip_atomic_store(wm + wave * 16 + l15, m_seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(wm + kQT + wave * 16 + l15, l_seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                                            // every wave's stores have landed
      if (tid == 0) __hip_atomic_fetch_add(pair_cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (owner_here) {
      int* done = pair_cnt + 1;
      if (tid == 0) {
        int polls = 0;
        bool gave_up = false;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nsp - 1) {
          __builtin_amdgcn_s_sleep(2);
          if (++polls > (1 << 21)) { gave_up = true; break; }
        }
        sgave = gave_up ? 1 : 0;
        if (gave_up) {   // (a workgroup of the pair never ran.  Never hang the GPU: count it in the time-out word, make the bank say
          atomicAdd(b.ovf + 1, 1);                                          // "do not trust me" and leave the counters alone; the
          atomicOr(b.ovf, kBankTimeout);                                    // launcher clears the control block before every read)
        } else {
          __hip_atomic_store(pair_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean counters for the next read
          __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();                                                              // E2: all partials of the pair are in memory
    } else {
      __syncthreads();                                                              // (Msh / Lsh visible)
    }
    // ---- merge (the last arriver of an in-place pair; or the only segment of the pair).  The others' partials were stored write-through and are
    //      read with sc1 loads (L2-coherent at agent scope: no acquire fence, no L1 invalidate needed).
    //      Consumers request the first foreign slot's fragments NOW, before the weights exist.
    int s_first = sself == 0 ? 1 : 0;                      // first foreign slot (if any)
    constexpr int kEarlyDt = kCDT / 2;                     // channel tiles of the early batch (registers: the whole slot spills)
    u32x4 v0[kEarlyDt][4];
    if (!producer && nsp > 1) {
      const __amdgpu_buffer_rsrc_t rs = slot_rsrc(ps.slot(s_first));
#pragma unroll
      for (int dt = 0; dt < kEarlyDt; ++dt)
#pragma unroll
        for (int it = 0; it < 4; ++it)
          v0[dt][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(partial_frag_offset(dt0 + dt, it, ln) * 4), 0, 16 /* sc1 */);
    }
    //      Producers: the pair's weights.  thread (sl = wave, qi = lane): query qi, slots sl, sl + 4, ...
    constexpr int kMl = 4;
    float m_r[kMl], l_r[kMl];
    auto ml_load = [&](int sj, int qi, float& m_, float& l_) {
      if (sj == sself) { m_ = Msh[qi]; l_ = Lsh[qi]; return; }
      const float* e = a.ws_ml + (size_t)ps.slot(sj) * 2 * kQT;
      m_ = __hip_atomic_load(e + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      l_ = __hip_atomic_load(e + kQT + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (producer) {
      const int qi = ln, sl = wave;
      float mloc = -INFINITY;
#pragma unroll
      for (int j = 0; j < kMl; ++j) {
        const int sj = sl + 4 * j;
        m_r[j] = -INFINITY; l_r[j] = 0.0f;
        if (sj < nsp) ml_load(sj, qi, m_r[j], l_r[j]);
        mloc = fmaxf(mloc, m_r[j]);
      }
      for (int sj = sl + 4 * kMl; sj < nsp; sj += 4) {
        float ms, ls;
        ml_load(sj, qi, ms, ls);
        mloc = fmaxf(mloc, ms);
      }
      red[sl * kQT + qi] = mloc;
    }
    __syncthreads();                                                                // E3
    float mtot_q = 0.0f;                                   // (producers) the pair's reference of query ln
    if (producer) {
      const int qi = ln, sl = wave;
      float mtot = fmaxf(fmaxf(red[qi], red[kQT + qi]), fmaxf(red[2 * kQT + qi], red[3 * kQT + qi]));
      if (n_out > 0.0f) mtot = fmaxf(mtot, 0.0f);          // the masked memory cells have S = 0
      mtot_q = mtot;
      float lloc = 0.0f;
#pragma unroll
      for (int j = 0; j < kMl; ++j) {
        const int sj = sl + 4 * j;
        const float wgt = __builtin_amdgcn_exp2f(m_r[j] - mtot);     // (-inf for a slot that does not exist: 0)
        if (sj < nsp) Wt[sj * kQT + qi] = wgt;
        lloc += l_r[j] * wgt;
      }
      for (int sj = sl + 4 * kMl; sj < nsp; sj += 4) {
        float ms, ls;
        ml_load(sj, qi, ms, ls);
        const float wgt = __builtin_amdgcn_exp2f(ms - mtot);
        Wt[sj * kQT + qi] = wgt;
        lloc += ls * wgt;
      }
      // (the closed-form term N_out * 2^(-m_tot) of the masked memory cells rides in group 0's partial sum)
      red2[sl * kQT + qi] = lloc + (sl == 0 && n_out > 0.0f ? n_out * __builtin_amdgcn_exp2f(-mtot) : 0.0f);
    }
    __syncthreads();                                                                // E4
    if (a.ml_out && wave == 0) {        // chained reads (> kMaxT frames): the state of this pair's queries
      const int nq = wk.qt * kQT + ln;
      if (nq < wk.Mq) {
        const int rw = wk.qr.width(), ry = nq / rw;
        const int cell = (wk.qr.cy0 + ry) * b.w + wk.qr.cx0 + (nq - ry * rw);
        float* ml = a.ml_out + (size_t)wk.o * 2 * hw;
        ml[cell] = mtot_q;
        ml[hw + cell] = red2[ln] + red2[kQT + ln] + red2[2 * kQT + ln] + red2[3 * kQT + ln];
      }
    }
    if (!producer) {
      // ---- consumers: O = (own x w_self + sum_s partial_s x w_s) / l_tot for this wave's 64 channels x 64 queries
      float wq[4], iq[4];
      // (a merge whose wait gave up -- cannot happen on a healthy device -- may not have all partials: its read-out is written as NaN,
      //  never as a half-merged value; the bank's sticky bit says "do not trust this read" as well)
      const float poison = sld(sgave) ? __builtin_nanf("") : kBankValueUnscale;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 16 + l15;
        wq[it] = Wt[sself * kQT + q];
        iq[it] = poison / (red2[q] + red2[kQT + q] + red2[2 * kQT + q] + red2[3 * kQT + q]);
      }
#pragma unroll
      for (int dt = 0; dt < kCDT; ++dt)
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[dt][it] *= wq[it];
      if (nsp > 1) {
        const __amdgpu_buffer_rsrc_t rs = slot_rsrc(ps.slot(s_first));
        u32x4 v1[kCDT - kEarlyDt][4];                      // the rest of the first foreign slot
#pragma unroll
        for (int dt = kEarlyDt; dt < kCDT; ++dt)
#pragma unroll
          for (int it = 0; it < 4; ++it)
            v1[dt - kEarlyDt][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(partial_frag_offset(dt0 + dt, it, ln) * 4), 0, 16 /* sc1 */);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const float ws_ = Wt[s_first * kQT + it * 16 + l15];
#pragma unroll
          for (int dt = 0; dt < kEarlyDt; ++dt) acc[dt][it] += ws_ * __builtin_bit_cast(f32x4, v0[dt][it]);
#pragma unroll
          for (int dt = kEarlyDt; dt < kCDT; ++dt) acc[dt][it] += ws_ * __builtin_bit_cast(f32x4, v1[dt - kEarlyDt][it]);
        }
      }
      for (int s_ = s_first + 1; s_ < nsp; ++s_) {
        if (s_ == sself) continue;
        const __amdgpu_buffer_rsrc_t rs = slot_rsrc(ps.slot(s_));
        u32x4 v[kCDT][4];
#pragma unroll
        for (int dt = 0; dt < kCDT; ++dt)
#pragma unroll
          for (int it = 0; it < 4; ++it)
            v[dt][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(partial_frag_offset(dt0 + dt, it, ln) * 4), 0, 16 /* sc1 */);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const float ws_ = Wt[s_ * kQT + it * 16 + l15];
#pragma unroll
          for (int dt = 0; dt < kCDT; ++dt) acc[dt][it] += ws_ * __builtin_bit_cast(f32x4, v[dt][it]);
        }
      }
      // ---- transpose 16 channels x 64 queries at a time through this wave's LDS patch and scatter to the cells:
      //      a store instruction covers 64 consecutive compacted queries of one channel row
      float* T = Tw + (wave - kProducers) * 16 * 65;
      const int nq = wk.qt * kQT + ln;
      const bool qvalid = nq < wk.Mq;
      int cell = 0;
      if (qvalid) {
        const int rw = wk.qr.width(), ry = nq / rw;
        cell = (wk.qr.cy0 + ry) * b.w + wk.qr.cx0 + (nq - ry * rw);
      }
      float* __restrict__ outo = a.out + (size_t)wk.o * 2 * kDo * hw + cell;
#pragma unroll
      for (int dt = 0; dt < kCDT; ++dt) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r) T[(4 * g + r) * 65 + it * 16 + l15] = acc[dt][it][r] * iq[it];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) {
          const float x = T[ch * 65 + ln];
          if (qvalid) outo[(size_t)(16 * (dt0 + dt) + ch) * hw] = x;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  };

  if (cl < nqt * bc.nfull) {          // aligned chunk: (column block, query tile), one segment
    const int blk = plan_div(cl, nqt);
    wk.qt = cl - blk * nqt;
    bank_block_range(bc, njt, blk, wk.jt0, wk.ntl);
    wk.slot = slot_obj + cl;
    wk.pf_part = wk.qt; wk.pf_nparts = nqt;
    run_segment(blk);
    return;
  }
  // remainder chunk: units [u0, u1) of the virtual line over the last R tile columns (common.h)
  const int cr = cl - nqt * bc.nfull;
  const int u0 = cr * C, u1 = u0 + C, span = bc.R + bc.sc;
  bool first = true;
  for (int qt = plan_div(u0, span); qt < nqt && qt * span < u1; ++qt) {
    const int j0 = max(u0 - qt * span, 0), j1 = min(u1 - qt * span, bc.R);   // tiles of pair qt inside the chunk
    if (j1 <= j0) continue;
    wk.qt = qt;
    wk.jt0 = bc.nfull * bc.Cb + j0;
    wk.ntl = j1 - j0;
    wk.slot = slot_obj + nqt * bc.nfull + cr + qt;
    wk.pf_part = 0; wk.pf_nparts = 0;          // (remainder chunks do not prefetch: alone they would touch every line of a step)
    if (!first) __syncthreads();      // the previous segment's LDS (K ring, P, alpha, epilogue scratch) is free
    first = false;
    run_segment(bc.nfull + cr - plan_div(qt * span, C));
  }
  };   // compute()
  if ((int)blockIdx.x < G) {
    int c0;
    {
      const int q8 = G >> 3, r8 = G & 7, x = blockIdx.x & 7;             // XCD-contiguous logical ids
      c0 = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (blockIdx.x >> 3);
    }
    for (int c = c0; c < nchunks; c += G) {                               // (one chunk unless the launch runs in rounds)
      if (c != c0) __syncthreads();   // the previous chunk's LDS (tile prefix, K ring, P, alpha, epilogue scratch) is free
      compute(c);
    }
    // [r6] the pairs this workgroup holds a segment of, now that its walks are over (the other workgroups of a pair are at most a
    // few microseconds behind: the chunks are equalised)
    __syncthreads();
    for (int c = c0; c < nchunks; c += G) merge_chunk(c);
  }

  // =========================================== static part: drain the queue ===========================================
  // A workgroup pulls tickets of 12 x kStaticRowsPerTicket (object, channel) rows (one atomic per ticket: with a
  // ticket per WAVE the 3,072 waves of a launch saturated the queue word -- ~88 atomics per us -- and a pull took
  // 35 us); inside a ticket every wave walks its own rows, no barrier.  A row is hw cells = hwv units of 16 bytes (4 bytes
  // when the rows cannot be moved 16 bytes at a time); lane l handles units l, l + 64, ...  The box test is per CELL,
  // identical for all 512 rows of an object: the wave keeps a byte per unit of its current object in its own LDS patch
  // (bits 0-3: cell inside the query box, bits 4-7: cell gets the mean) and a 16-entry table turns the low nibble into a
  // 0/1 float4.  Everything goes through buffer instructions whose offset is pushed out of range for the lanes that must
  // not store: STRAIGHT-LINE code.  (With branches around the stores hipcc could not count the loads in flight and put
  // s_waitcnt vmcnt(0) in front of every unit, i.e. every unit waited for the previous unit's stores to land: 12-14 us
  // per ticket whatever else was tried.)
  {
    const bool vec4 = (hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.qv) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0;
    constexpr int kMaskUnits = 4096;                   // units of a wave's mask patch (longer rows are walked in pieces)
    constexpr int kUI = 7;                             // units per lane and step (7 x 64 = 448 >= the 405 units of a 480p row)
    constexpr int kMixMax = 512;                       // mixed units (some cells of the unit get the mean, some do not) per piece
    constexpr int kOOB = 0x40000000;                   // a byte offset no row reaches: the buffer unit drops the access
    constexpr int kRowsPerTicket = kStaticRowsPerTicket * (kProducers + kConsumers);
    const int nrow_total = ng * kDo;
    const int ntickets = (nrow_total + kRowsPerTicket - 1) / kRowsPerTicket;
    const float* __restrict__ qv0 = a.qv + (size_t)a.obj0 * kDo * hw;      // rows (og, d) of the launch, contiguous
    float* __restrict__ out0 = a.out + (size_t)a.obj0 * 2 * kDo * hw;
    const float inv_cells = 1.0f / ((float)T_ * (float)hw);
    unsigned char* umask = reinterpret_cast<unsigned char*>(Kl_) + wave * kMaskUnits;      // this wave's patch
    f32x4* lut = reinterpret_cast<f32x4*>(Kl_ + (kProducers + kConsumers) * kMaskUnits);   // [16] nibble -> 0/1 float4
    unsigned short* mixlist = reinterpret_cast<unsigned short*>(Pl_) + wave * kMixMax;     // this wave's mixed units: unit | nibble << 12
    int nmix = 0, mask_obj = -1, mask_piece = -1;      // what this wave's patch currently holds
    auto row_rsrc = [&](const float* base) {           // descriptor of one row (wave-uniform base, hw floats)
      const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(base));
      const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(base) >> 32));
      return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uintptr_t)bhi << 32) | blo), 0, hw * 4, 0x00020000);
    };
    auto build_masks = [&](int og, int p0, int pn, int nvec) {
      const Rect rc{o_rect[og][0], o_rect[og][1], o_rect[og][2], o_rect[og][3]};
      const bool nomem = o_njt[og] == 0;
      const unsigned full = nvec == 4 ? 15u : 1u;
      nmix = 0;
      for (int cb = 0; cb < pn; cb += RMNET_WAVE) {
        const int cu = cb + lane0;
        int cell = (p0 + cu) * nvec, cy = cell / b.w, cx = cell - cy * b.w;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool inside = e < nvec && rc.contains(cy, cx);
          m |= (inside ? 1u : 0u) << e;
          m |= ((!inside || nomem) && e < nvec ? 1u : 0u) << (4 + e);
          if (++cx == b.w) { cx = 0; ++cy; }
        }
        const bool mixed = cu < pn && (m >> 4) != 0u && (m >> 4) != full;   // handled by the per-row list pass
        const unsigned long long bal = __ballot(mixed);
        if (mixed) {
          const int at = nmix + __popcll(bal & ((1ull << lane0) - 1ull));
          if (at < kMixMax) mixlist[at] = (unsigned short)(cu | (m >> 4) << 12);
        }
        nmix += __popcll(bal);
        if (cu < pn) umask[cu] = (unsigned char)m;
      }
      __builtin_amdgcn_wave_barrier();
    };
    // One step = kUI units per lane of one row.  VEC4: 16-byte units, else 4-byte units (same code, narrower accesses).
    auto run_ticket = [&](auto vec_tag, int ticket) {
      constexpr bool V4 = decltype(vec_tag)::value;
      constexpr int nvec = V4 ? 4 : 1, ub_bytes = V4 ? 16 : 4;
      const int hwv = V4 ? hw >> 2 : hw;
      const int row0w = ticket * kRowsPerTicket + wave * kStaticRowsPerTicket;
      // column sums of this wave's rows (lanes over the frames, fixed order): requested first, used row by row
      float mus[kStaticRowsPerTicket];
#pragma unroll
      for (int rr = 0; rr < kStaticRowsPerTicket; ++rr) {
        const int row = min(row0w + rr, nrow_total - 1), og = row / kDo, d = row - og * kDo;
        const float* __restrict__ cs = b.colsum + ((size_t)(a.obj0 + og) * b.Tcap) * kDo + d;
        mus[rr] = 0.0f;
        for (int t = lane0; t < T_; t += RMNET_WAVE) mus[rr] += cs[(size_t)t * kDo];
      }
      // ... and so is the first step of every row: ALL of the ticket's loads are in flight before the first store
      // (a row-by-row walk pays one full memory latency per row: measured 5 us per row and wave)
      u32x4 x0[kStaticRowsPerTicket][kUI];
#pragma unroll
      for (int rr = 0; rr < kStaticRowsPerTicket; ++rr) {
        const int row = min(row0w + rr, nrow_total - 1);
        const __amdgpu_buffer_rsrc_t rs_src = row_rsrc(qv0 + (size_t)row * hw);
#pragma unroll
        for (int i = 0; i < kUI; ++i) {
          const int off = (i * RMNET_WAVE + lane0) * ub_bytes;
          if (V4) x0[rr][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0);
          else x0[rr][i] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(rs_src, off, 0, 0), 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int rr = 0; rr < kStaticRowsPerTicket; ++rr) {
        const int row = row0w + rr;
        if (row >= nrow_total) break;
        const int og = row / kDo, d = row - og * kDo;
        const float mu = wave_sum_f(mus[rr]) * inv_cells;     // the row's mean of m_val over ALL T*h*w cells
        const u32x4 mu4 = {__float_as_uint(mu), __float_as_uint(mu), __float_as_uint(mu), __float_as_uint(mu)};
        const __amdgpu_buffer_rsrc_t rs_src = row_rsrc(qv0 + (size_t)row * hw);
        const __amdgpu_buffer_rsrc_t rs_q = row_rsrc(out0 + ((size_t)og * 2 * kDo + kDo + d) * hw);   // out[og][kDo + d]
        const __amdgpu_buffer_rsrc_t rs_m = row_rsrc(out0 + ((size_t)og * 2 * kDo + d) * hw);         // out[og][d]
        for (int piece = 0; piece * kMaskUnits < hwv; ++piece) {
          const int p0 = piece * kMaskUnits, pn = min(hwv - p0, kMaskUnits);
          if (og != mask_obj || piece != mask_piece) {
            build_masks(og, p0, pn, nvec);
            mask_obj = og; mask_piece = piece;
          }
          for (int ub = 0; ub < pn; ub += kUI * RMNET_WAVE) {
            u32x4 x[kUI];
            if (ub == 0 && piece == 0) {
#pragma unroll
              for (int i = 0; i < kUI; ++i) x[i] = x0[rr][i];
            } else {                                   // (rows longer than one step: later steps are loaded here)
#pragma unroll
              for (int i = 0; i < kUI; ++i) {          // (a unit past the row: out of range, returns 0)
                const int off = (p0 + ub + i * RMNET_WAVE + lane0) * ub_bytes;
                if (V4) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, off, 0, 0);
                else x[i] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(rs_src, off, 0, 0), 0u, 0u, 0u};
              }
            }
#pragma unroll
            for (int i = 0; i < kUI; ++i) {
              const int ul = ub + i * RMNET_WAVE + lane0;                    // unit inside the piece
              const int off = (p0 + ul) * ub_bytes;
              const unsigned m = umask[min(ul, kMaskUnits - 1)];             // (past the piece: whatever -- the stores are dropped)
              const f32x4 keep = lut[m & 15u];
              f32x4 y = __builtin_bit_cast(f32x4, x[i]);
#pragma unroll
              for (int e = 0; e < nvec; ++e) y[e] *= keep[e];              // x * 1 = x, x * 0 = +-0 / NaN: the reference's q_val * box (:358)
              const int qoff = ul < pn ? off : kOOB;
              const int moff = (ul < pn && (m >> 4) == (V4 ? 15u : 1u)) ? off : kOOB;   // (mixed units: the list pass)
              if (V4) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rs_q, qoff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(mu4, rs_m, moff, 0, 0);
              } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[0]), rs_q, qoff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mu), rs_m, moff, 0, 0);
              }
            }
          }
          // ---- the few units with cells on both sides of the box edge: one lane per unit, per-cell stores
          if (V4) {
            if (nmix > kMixMax) {                                          // (cannot happen on RMNet's grids: walk all units)
              for (int cu = lane0; cu < pn; cu += RMNET_WAVE) {
                const unsigned nib = umask[cu] >> 4;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mu), rs_m,
                                                        (nib != 15u && (nib >> e & 1u)) ? (p0 + cu) * 16 + e * 4 : kOOB, 0, 0);
              }
            } else {
              for (int k = lane0; k < nmix; k += RMNET_WAVE) {
                const unsigned ent = mixlist[k];
                const int off = (p0 + (int)(ent & 0xfffu)) * 16;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mu), rs_m, (ent >> (12 + e) & 1u) ? off + e * 4 : kOOB, 0, 0);
              }
            }
          }
        }
      }
    };
    // Inside the loop only LDS (the ticket word) crosses the barrier: a plain __syncthreads() is also a memory fence and
    // waits for every outstanding global store of the wave, so the loop uses the bare barrier behind an LDS-only wait.
    auto lds_barrier = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    __syncthreads();                                   // (the compute part's LDS traffic is over)
    if (tid < 16) lut[tid] = f32x4{(tid & 1) ? 1.f : 0.f, (tid & 2) ? 1.f : 0.f, (tid & 4) ? 1.f : 0.f, (tid & 8) ? 1.f : 0.f};
    // (a LOAD first: when the ~200 compute workgroups of a launch finish within a few microseconds of each other the queue is
    //  usually empty already, and 200 returning atomics on one word serialise at ~88 per microsecond -- the last workgroup left
    //  8 us after its epilogue without having served a ticket; loads of the same word do not queue up)
    if (tid == 0) {
      const int h = __hip_atomic_load(q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sflag = h >= ntickets ? h : __hip_atomic_fetch_add(q_head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    int ticket = sld(sflag);
    while (ticket < ntickets) {
      lds_barrier();                                   // (everybody has read sflag)
      int t_next = 0;                                  // the next ticket flies while this one is worked on
      if (tid == 0) t_next = __hip_atomic_fetch_add(q_head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (vec4) run_ticket(std::true_type{}, ticket);
      else run_ticket(std::false_type{}, ticket);
      if (tid == 0) sflag = t_next;
      lds_barrier();
      ticket = sld(sflag);
    }
  }
  // (the queue word is NOT reset here: the launcher clears the control block before every launch -- a last-one-out counter
  //  was another 256 same-word atomics at the very end of the kernel)
}

}  // namespace

// ---- reads of more than kMaxT memorised frames (models/rmnet.py:416-426 has no bound): the launcher reads the bank in
// chunks of <= kMaxT slots -- every chunk is an ordinary bk_main launch that also leaves the soft-max state (m, l) of its
// query cells -- and bk_chain merges the chunks' read-outs exactly as the last arriver merges the partials of a pair:
//     out = sum_c w_c out_c / sum_c w_c,   w_c = l_c 2^(m_c - max_c m_c).
// A cell that went through no merge (outside the query box, or no memory cell inside any box of the chunk) was written by
// the static part with the mean of the chunk's values: all its logits are 0, i.e. m = 0 and l = T_c h w -- the state the
// launcher pre-fills (bk_ml_fill).
namespace {
__global__ __launch_bounds__(256) void bk_ml_fill(float* ml, int no, int hw, float l0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < no * 2 * hw) ml[i] = ((i / hw) & 1) ? l0 : 0.0f;
}
constexpr int kChainMax = 64;
struct ChainArgs { float* out; const float* tmp; const float* ml; int no, hw, nchunk; };
// grid (ceil(hw / 256), no): thread = one query cell, all 512 read-out channels (coalesced over the cells)
__global__ __launch_bounds__(256) void bk_chain(const ChainArgs c) {
  const int cell = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  if (cell >= c.hw) return;
  float wgt[kChainMax];
  float mmax = -INFINITY;
  for (int k = 0; k < c.nchunk; ++k) mmax = fmaxf(mmax, c.ml[((size_t)(k * c.no + o) * 2) * c.hw + cell]);
  float lsum = 0.0f;
  for (int k = 0; k < c.nchunk; ++k) {
    const float* ml = c.ml + ((size_t)(k * c.no + o) * 2) * c.hw;
    wgt[k] = ml[c.hw + cell] * __builtin_amdgcn_exp2f(ml[cell] - mmax);
    lsum += wgt[k];
  }
  const float inv = 1.0f / lsum;
  float* dst = c.out + (size_t)o * 2 * kDo * c.hw + cell;
  for (int d = 0; d < kDo; ++d) {
    float acc = wgt[0] * dst[(size_t)d * c.hw];
    for (int k = 1; k < c.nchunk; ++k)
      acc += wgt[k] * c.tmp[((size_t)((k - 1) * c.no + o) * 2 * kDo + d) * c.hw + cell];   // (a chunk's whole [no][2 Do][hw] output)
    dst[(size_t)d * c.hw] = acc * inv;
  }
}
}  // namespace

// Clears `nwords` int32 of a bank's control block.  A KERNEL, not hipMemsetAsync: as a memset node of a captured HIP graph the
// runtime's fill faulted on the second replay about every other run (MI355X, ROCm 7.2: `Memory access fault by GPU`, found by
// bench.py's graph section in round 4; `tools/repro_single.py GRAPH=1`), a kernel node of the same graph never did.
namespace {
__global__ __launch_bounds__(256) void bk_ctl_clear(int32_t* w, int nwords) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nwords; i += gridDim.x * 256) w[i] = 0;
}
}  // namespace
int launch_bank_ctl_clear(int32_t* words, int nwords, hipStream_t st) {
  if (nwords <= 0) return RMNET_OK;
  hipLaunchKernelGGL(bk_ctl_clear, dim3(nwords > 4096 ? 8 : 1), dim3(256), 0, st, words, nwords);
  return check_launch();
}

int bank_chain_max_chunks() { return kChainMax; }
int bank_max_frames_per_launch() { return kMaxT; }

int launch_bank_ml_fill(float* ml, int no, int hw, float l0, hipStream_t st) {
  hipLaunchKernelGGL(bk_ml_fill, dim3((no * 2 * hw + 255) / 256), dim3(256), 0, st, ml, no, hw, l0);
  return check_launch();
}

int launch_bank_chain(float* out, const float* tmp, const float* ml, int no, int hw, int nchunk, hipStream_t st) {
  if (nchunk < 2 || nchunk > kChainMax) return RMNET_E_UNSUPPORTED;
  ChainArgs c{out, tmp, ml, no, hw, nchunk};
  hipLaunchKernelGGL(bk_chain, dim3((hw + 255) / 256, no), dim3(256), 0, st, c);
  return check_launch();
}

int launch_bank_append(void* bank, int no, int Tcap, int h, int w, int slot, const float* k4,
                       const float* v4, const int32_t* rects, hipStream_t st, const int32_t* slot_dev) {
  const long long hw = (long long)h * w;
  return launch_bank_stage(bank, no, Tcap, h, w, slot, 1, k4, v4, hw, hw * kDe, hw, hw * kDo, rects, st, slot_dev);
}

int launch_bank_stage(void* bank, int no, int Tcap, int h, int w, int slot0, int nf, const float* k4,
                      const float* v4, long long k_cs, long long k_os, long long v_cs, long long v_os,
                      const int32_t* rects, hipStream_t st, const int32_t* slot_dev, bool colsum) {
  if (!bank || !k4 || !v4 || no <= 0 || Tcap <= 0 || h <= 0 || w <= 0 || nf <= 0 || slot0 < 0 ||
      slot0 + nf > Tcap)
    return RMNET_E_INVALID_ARG;
  if ((long long)no * nf > 65535) return RMNET_E_UNSUPPORTED;
  const BankView b = bank_view(bank, no, Tcap, h, w);
  hipLaunchKernelGGL(bk_append, dim3(b.hwp / kJT, no * nf, 1 + kDo / kDe), dim3(kThreads), 0, st, b, slot0, slot_dev, nf,
                     k4, v4, k_cs, k_os, v_cs, v_os, rects);
  if (int e = check_launch()) return e;
  // the slots' column sums feed the read-out of query cells OUTSIDE the query box (and of objects without a memory cell): a dense
  // read (the drop-in entry without rectangles) has neither and skips the launch
  if (!colsum) return RMNET_OK;
  hipLaunchKernelGGL(bk_colsum, dim3(no * nf), dim3(kDo), 0, st, b, slot0, slot_dev, nf);
  return check_launch();
}

size_t bank_overflow_offset(int no, int Tcap, int h, int w) { return bank_bytes(no, Tcap, h, w) - bank_ctl_bytes(no, h, w); }

int launch_bank_main(const BankReadArgs& m, hipStream_t st) {
  BArgs a;
  a.b = bank_view(m.bank, m.no, m.Tcap, m.h, m.w);
  if (m.t0 < 0 || m.t0 >= m.Tcap) return RMNET_E_INVALID_ARG;
  if (m.t0 > 0) {   // a read of slots [t0, ...): the same view with every per-slot array advanced by t0 slots (object stride = Tcap)
    a.b.kh += (size_t)m.t0 * a.b.hwp * kDe * sizeof(_Float16);
    a.b.kl += (size_t)m.t0 * a.b.hwp * kDe * sizeof(_Float16);
    a.b.vh += (size_t)m.t0 * kDo * a.b.hwp * sizeof(_Float16);
    a.b.vl += (size_t)m.t0 * kDo * a.b.hwp * sizeof(_Float16);
    a.b.vpart += (size_t)m.t0 * (a.b.hwp / kJT) * kDo;
    a.b.colsum += (size_t)m.t0 * kDo;
    a.b.area += m.t0;
  }
  a.Tmax = m.Tcap - m.t0 < kMaxT ? m.Tcap - m.t0 : kMaxT;
  a.ml_out = m.ml_out;
  a.qk = m.qk; a.qv = m.qv; a.qry_rects = m.qry_rects;
  a.out = m.out;
  a.ws_o = m.ws_o; a.ws_ml = m.ws_ml; a.ws_plan = m.ws_plan;
  a.T = m.T;
  a.T_dev = m.T_dev;
  a.gate = m.gate;
  a.qscale = 1.44269504088896341f / sqrtf((float)kDe) * kBankScale;   // (the un-scaling is kSraw in the soft-max)
  // Objects are planned together in groups of <= kMaxObj; a group's partial slots start at
  // bank_group_slot0() and hold at most target + nobj * (query tiles) segments.
  for (int obj0 = 0; obj0 < m.no; obj0 += kMaxObj) {
    a.obj0 = obj0;
    a.nobj = m.no - obj0 < kMaxObj ? m.no - obj0 : kMaxObj;
    a.slot0 = bank_group_slot0(obj0, m.h * m.w);
    a.target = kSplitTargetSlots;
    // the static work queue's word must be zero at every launch: the caller (launch_bank_read / the drop-in entry) has cleared
    // the control block for the first group, the later groups clear it here
    if (obj0 > 0)
      if (int e = launch_bank_ctl_clear(a.b.ovf + 16, 16, st)) return e;
    if (m.f16 == 1)
      hipLaunchKernelGGL(bk_main<1>, dim3(a.target), dim3(kRThreads), 0, st, a);
    else if (m.f16 == 2)
      hipLaunchKernelGGL(bk_main<2>, dim3(a.target), dim3(kRThreads), 0, st, a);
    else
      hipLaunchKernelGGL(bk_main<3>, dim3(a.target), dim3(kRThreads), 0, st, a);
    if (int e = check_launch()) return e;
  }
  return RMNET_OK;
}

}  // namespace rmnet
