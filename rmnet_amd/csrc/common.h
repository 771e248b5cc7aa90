// Shared device helpers for librmnet_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rmnet_hip.h"

#define RMNET_WAVE 64

// gfx950 only: the read kernels size their static LDS (one workgroup per CU) against CDNA4's 160 KB; on the 64 KB parts
// (gfx90a / gfx942) they cannot be launched at all.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "librmnet_hip is written for gfx950 (MI355X): build with --offload-arch=gfx950"
#endif
constexpr int kLdsBytesPerCU = 160 * 1024;

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

#if defined(__HIPCC__)
// Wave-wide integer reductions / scan on the VALU's data-parallel primitives (DPP row operations inside a row of 16 lanes, then
// v_permlane16_swap / v_permlane32_swap across the four rows) instead of __shfl_*: hipcc lowers a shuffle to ds_bpermute, i.e.
// one LDS crossbar round trip (~100+ cycles) per step, and the launch plan of bk_main runs a dozen of them back to back on the
// critical path of every workgroup.
#define RMNET_DPP(old, src, ctrl, rmask, bmask, bc) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), (bmask), (bc))
__device__ inline int wave_sum_fast(int v) {          // every lane gets the sum over the 64 lanes
  v += RMNET_DPP(0, v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
  v += RMNET_DPP(0, v, 0x4E, 0xf, 0xf, false);        // quad_perm [2,3,0,1]: every lane = its quad's sum
  v += RMNET_DPP(0, v, 0x141, 0xf, 0xf, false);       // row_half_mirror: + the other quad of the half row
  v += RMNET_DPP(0, v, 0x140, 0xf, 0xf, false);       // row_mirror: + the other half of the row
  auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = (int)r[0] + (int)r[1];
  r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)r[0] + (int)r[1];
}
__device__ inline int wave_max_fast(int v) {
  v = max(v, RMNET_DPP(v, v, 0xB1, 0xf, 0xf, false));
  v = max(v, RMNET_DPP(v, v, 0x4E, 0xf, 0xf, false));
  v = max(v, RMNET_DPP(v, v, 0x141, 0xf, 0xf, false));
  v = max(v, RMNET_DPP(v, v, 0x140, 0xf, 0xf, false));
  auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = max((int)r[0], (int)r[1]);
  r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return max((int)r[0], (int)r[1]);
}
__device__ inline int wave_scan_incl_fast(int v) {    // lane l gets v[0] + ... + v[l]
  v += RMNET_DPP(0, v, 0x111, 0xf, 0xf, true);        // row_shr:1 (lanes shifted in from outside the row read 0)
  v += RMNET_DPP(0, v, 0x112, 0xf, 0xf, true);        // row_shr:2
  v += RMNET_DPP(0, v, 0x114, 0xf, 0xf, true);        // row_shr:4
  v += RMNET_DPP(0, v, 0x118, 0xf, 0xf, true);        // row_shr:8  -> inclusive scan inside every row of 16
  v += RMNET_DPP(0, v, 0x142, 0xa, 0xf, false);       // row_bcast15 into rows 1 and 3: + the total of the row before
  v += RMNET_DPP(0, v, 0x143, 0xc, 0xf, false);       // row_bcast31 into rows 2 and 3: + the total of the first 32 lanes
  return v;
}
#endif

// Launchers (one per .hip file); each returns RMNET_OK or a negative code.
int launch_region_map_warped(const float* mask, const float* flow, int B, int K, int H, int W,
                             float thr, int n_pts, int loose, float* att, int32_t* bboxes,
                             int32_t* cell_rects, int pad_l, int pad_t, int cell_stride, int ch,
                             int cw, float* warped, void* ws, size_t ws_bytes, hipStream_t st);
int launch_region_map(const float* mask, int B, int K, int H, int W, float thr, int n_pts,
                      int loose, float* att, int32_t* bboxes, int32_t* cell_rects, int pad_l,
                      int pad_t, int cell_stride, int ch, int cw, void* ws, size_t ws_bytes,
                      hipStream_t st);
size_t region_map_ws_bytes(int B, int K, int H, int W);
int launch_boxes_to_rects(const int32_t* bboxes, int n, int k_per_batch, int pad_l, int pad_t,
                          int stride, int ch, int cw, int32_t* rects, hipStream_t st);
int launch_rect_mask(const float* x, int n, int C, int T, int h, int w, const int32_t* rects,
                     float* y, hipStream_t st);
int launch_channel_affine(const float* x, const float* scale, const float* shift, const float* res,
                          const float* rscale, const float* rshift, int relu, long long N, int C,
                          long long HW, float* out, hipStream_t st);
int launch_upsample2x_add(const float* x, const float* skip, long long N, int C, int h, int w,
                          float* out, hipStream_t st);
int launch_soft_aggregate(const float* dec, const int32_t* obj_begin, int B, int K, int Hp, int Wp,
                          int pad_l, int pad_t, int H, int W, float* logit, float* prob,
                          hipStream_t st);
int launch_affine_relu_maxpool(const float* x, const float* scale, const float* shift, long long N,
                               int C, int H, int W, float* out, hipStream_t st);
int launch_flow_affine(const float* flow, const float* m1, const float* m2, int H, int W,
                       float* out, hipStream_t st);
// channels-last ([N, H, W, C] in memory) variants of the three glue passes (epilogue.hip)
int launch_channel_affine_nhwc(const float* x, const float* scale, const float* shift, const float* res, const float* rscale,
                               const float* rshift, int relu, long long rows, int C, float* out, hipStream_t st);
int launch_upsample2x_add_nhwc(const float* x, const float* skip, long long N, int C, int h, int w, float* out, hipStream_t st);
int launch_affine_relu_maxpool_nhwc(const float* x, const float* scale, const float* shift, long long N, int C, int H, int W, float* out,
                                    hipStream_t st);

// Split partials in the workspace: per (object, slot) a [32 channel tiles][4 query tiles][64 lanes][4]
// fp32 block = the 16x16 MFMA accumulator fragments exactly as they sit in registers
// (lane = 16 * ((channel % 16) / 4) + query % 16, element = channel % 4).  A wave stores 1 KB
// contiguous per fragment and the combine kernel reads whole fragments back (16 B per lane).
__host__ __device__ inline size_t partial_frag_offset(int dtile, int qtile, int lane) {
  return ((size_t)(dtile * 4 + qtile) * 64 + lane) * 4;
}
__host__ __device__ inline size_t partial_elem_offset(int query, int channel) {
  return partial_frag_offset(channel >> 4, query >> 4, 16 * ((channel & 15) >> 2) + (query & 15)) + (channel & 3);
}

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

// ---- split-fp16 memory bank (bank.hip) -------------------------------------------------------
struct BankView {
  char *kh, *kl;   // [no][Tcap][hwp][128] fp16 hi / lo  (cell-major keys)
  char *vh, *vl;   // [no][Tcap][512][hwp] fp16 hi / lo  (channel-major values, cells permuted per 32)
  float* vpart;    // [no][Tcap][hwp/32][512] fp32: per 32-cell tile, the sum of the (un-scaled) values of its cells
  float* colsum;   // [no][Tcap][512] fp32: sum of a slot's values over the cells inside its box (bk_colsum)
  int32_t* area;   // [no][Tcap] cells inside the box of each memorised frame
  int32_t* ovf;    // control block (256 B): [0] number of 16-byte groups written (or query elements read) so far that held an element
                   // outside fp16's window, plus the sticky error bits 1 << 30 (slot / frame count out of range) and 1 << 29 (a merge
                   // timed out); [1] merges that timed out; [2] largest soft-max reference of any read so far (float bits, log2 domain, >= 0); [16] the read kernel's static work queue (next item).  Bytes 64.. (queue words, arrival counters) are cleared by the launcher before every read
  int32_t* cnt;    // [no][nqt_max][2] (merged, published) counters of the partials of an (object, query tile) pair
                   // (all queue words and counters are zero between reads)
  int no, Tcap, h, w, hw, hwp;
};
BankView bank_view(void* base, int no, int Tcap, int h, int w);
size_t bank_bytes(int no, int Tcap, int h, int w);
size_t bank_area_offset(int no, int Tcap, int h, int w);
size_t bank_ctl_bytes(int no, int h, int w);      // overflow word + arrival tickets (zero in a new bank)
constexpr float kBankValueUnscale = 1.0f / 64.0f;   // values are stored times 2^6 (bank.hip)

// Split heuristic shared by the read kernels and the combine kernel (must agree exactly).
#ifndef RMNET_SPLIT_TARGET
#define RMNET_SPLIT_TARGET 256
#endif
constexpr int kSplitTargetSlots = RMNET_SPLIT_TARGET;   // workgroups per launch to aim for (one per CU)
#ifndef RMNET_SPLIT_MIN_TILES
#define RMNET_SPLIT_MIN_TILES 4   // measured, one 480p object: box 19 % of the cells 38.8 -> 32.7 us (read + combine), 5 %: 28.1 -> 29.7
#endif
constexpr int kSplitMinTiles = RMNET_SPLIT_MIN_TILES;   // a split must amortise its prologue + 128 KB partial
constexpr int kSplitMax = 64;
struct BankPlan { int nqt, nsplit; };
__host__ __device__ inline BankPlan bank_plan(int Mq, int hw, int njt, int no, int slots) {
  BankPlan p;
  p.nqt = (Mq + (Mq < hw ? 1 : 0) + 63) / 64;   // +1: the mean slot for masked query cells
  if (p.nqt < 1) p.nqt = 1;
  int ns = kSplitTargetSlots / (p.nqt * no);
  if (ns > slots / p.nqt) ns = slots / p.nqt;
  if (ns > kSplitMax) ns = kSplitMax;
  if (ns > njt / kSplitMinTiles) ns = njt / kSplitMinTiles;
  if (ns < 1) ns = 1;
  if (ns > njt) ns = njt;   // 0 when there is nothing to read
  p.nsplit = ns;
  return p;
}

// Plan record, 12 ints per object.  mr_main writes it for mr_combine (one small load instead of re-deriving the
// plan from the rectangles): {Mq, nqt, nsplit, M, qr.cx0, qr.cx1, qr.cy0, qr.cy1, first partial slot of the
// object, 0, 0, 0}; the partial of (split s, query tile qt) is slot first + s * nqt + qt.  bk_main merges its
// partials itself and leaves {Mq, nqt, njt, M, qr..., first partial slot, chunk length C, 0, 1} for tools only.
constexpr int kPlanInts = 12;

// Chunking of one object's nqt x njt tile matrix into workgroup chunks of cost C (bank.hip):
// nfull column blocks of exactly C tiles (one single-segment chunk per (block, query tile)), then
// the remaining R tile columns, query-tile-major: pair qt occupies [qt * (R + kSegCost), + R) of a
// virtual line that is cut every C units -- the kSegCost units after a pair's tiles are empty and
// stand for the price of starting another segment (query fragments, K ring fill, first soft-max, a
// 128 KB partial), so a chunk that crosses pair boundaries gets that many tiles less.
#ifndef RMNET_SEG_COST
#define RMNET_SEG_COST 6   // (4 until round 3: a segment now also ends with a publish / ticket / merge)
#endif
constexpr int kSegCost = RMNET_SEG_COST;
#ifndef RMNET_SEG_COST_F16
#define RMNET_SEG_COST_F16 12   // fp16-operand mode: a tile costs a third, a segment's fixed part does not
#endif
constexpr int kSegCostF16 = RMNET_SEG_COST_F16;
// kTerms = the arithmetic of a bank read: 3 = split fp16 (three MFMA terms everywhere), 1 = fp16 operands, 2 = fp16 operands with an
// exact query (two terms for the logits, one for O = V P)
__host__ __device__ constexpr int seg_cost_of(int kTerms) { return kTerms != 3 ? kSegCostF16 : kSegCost; }
// Division of the launch plan's small non-negative integers (all < 2^22).  On the device hipcc expands an integer division by a
// run-time divisor into ~35 instructions, and the plan + the chunk lookup of bk_main do a dozen of them on the critical path of
// every workgroup; an fp32 reciprocal is off by at most one there, and the remainder fixes it up (exact, ~8 instructions).
__host__ __device__ inline int plan_div(int x, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
  int q = (int)((float)x * __builtin_amdgcn_rcpf((float)c));
  const int r = x - q * c;
  return q + (r >= c ? 1 : 0) - (r < 0 ? 1 : 0);
#else
  return x / c;
#endif
}
struct BankChunks { int C, Cb, nfull, R, nrem, nch, sc, eq, bq, br; };
// `blocks` is a launch-wide decision of the plan (bank.hip):
//   0  plain: nfull = njt / C aligned column blocks of C tiles + remainder chunks;
//   1  "own blocks": as 0, but an object with no more tiles than a chunk is ONE column block of its own length (see below);
//   2  EQUALISED: every object is cut into nfull = ceil(njt / C) column blocks of (almost) equal length <= C -- the first
//      njt mod nfull blocks have one tile more than the others -- and there are NO remainder chunks: every chunk is one segment, a pair has
//      exactly nfull partials.  The plan takes it when it fits the workgroups at (nearly) the chunk length the plain plan found:
//      the bench launch (objects of 119 tiles, C = 58) then walks 60 + 59 tiles with two partials per pair instead of
//      58 + 58 + 3 with three, and the multi-segment remainder chunks (a segment's fixed cost is ~12 tiles of the fp16 walk) are gone.
__host__ __device__ inline BankChunks bank_chunks(int nqt, int njt, int C, int segcost = kSegCost, int blocks = 0) {
  BankChunks k;
  k.C = C;
  k.sc = segcost;
  k.eq = 0;
  k.bq = k.br = 0;
  k.Cb = C;                          // length of an aligned column block
  if (blocks == 2) {
    k.eq = 1;
    k.nfull = njt > 0 ? plan_div(njt + C - 1, C) : 0;
    k.bq = k.nfull > 0 ? plan_div(njt, k.nfull) : 0;              // every block has bq tiles, the first br of them one more
    k.br = njt - k.bq * k.nfull;                                  // (one division of small operands: plan_div's < 2^22 contract holds)
    k.Cb = k.bq + (k.br > 0 ? 1 : 0);                             // (the longest block; bank_block_range() has each block's own)
    k.R = 0;
    k.nrem = 0;
    k.nch = nqt * k.nfull;
    return k;
  }
  k.nfull = plan_div(njt, C);
  k.R = njt - k.nfull * C;
  // An object with no more tiles than a chunk is ONE column block of its own length: nqt single-segment chunks that
  // walk the same tiles in lockstep.  As a "remainder" its pairs would be cut at shifted positions, every workgroup
  // would stream its own copy of the tiles, and two such objects per XCD do not fit the L2 (measured, 16 objects of
  // 120 tiles at C = 122: 164 us against 118 us for 15 objects at C = 108).
  // C is found WITHOUT it (the chunk count then falls as C grows, so a launch with more pairs than workgroups still
  // fits); it is switched on if the launch still fits with it.
  if (blocks == 1 && njt > 0 && njt <= C) { k.Cb = njt; k.nfull = 1; k.R = 0; }
  k.nrem = k.R > 0 && nqt > 0 ? plan_div(nqt * (k.R + segcost) - segcost + C - 1, C) : 0;   // (no query tile: no chunk)
  k.nch = nqt * k.nfull + k.nrem;
  return k;
}
// Tiles [j0, j0 + n) of aligned column block `blk` of an object with njt tiles.
__host__ __device__ inline void bank_block_range(const BankChunks& k, int njt, int blk, int& j0, int& n) {
  if (k.eq) {   // (round-5 advisor: floor(blk njt / nfull) took plan_div's numerator to nfull x njt, beyond its documented 2^22 for long banks)
    j0 = blk * k.bq + (blk < k.br ? blk : k.br);
    n = k.bq + (blk < k.br ? 1 : 0);
    (void)njt;
  } else {
    j0 = blk * k.Cb;
    n = k.Cb;
  }
}
// [r6] ROUNDS: a launch with more (object, query tile) pairs than workgroups.  Cutting the work evenly there means chunks that cross
// pair boundaries at shifted positions -- every workgroup streams its own copy of the K / V tiles and an XCD's L2 holds four objects
// at once (measured, 20 / 24 / 32 object-frames: 0.39 / 0.42 / 0.49 of the roof against 0.65 at 16).  Instead the workgroups run
// ROUNDS of aligned chunks: with P pairs on G workgroups, the first objects -- as many as fill R = P / G whole rounds -- are walked as
// ONE column block each (nqt workgroups in lockstep on one XCD, no partial results, no merge); the objects behind them are cut into b
// equal column blocks so that their chunks fill (at most) one more round.  Chunk c of the launch runs in round c / G on workgroup
// c % G.  The chunk length of the object whose inclusive pair prefix is ps (pairs of this and all earlier objects):
__host__ __device__ inline int bank_round_chunk_len(int njt, int ps, int P, int Pw, int G, int cq) {
  const int R = plan_div(P, G);
  if (njt <= 0 || ps <= R * G) return njt > 0 ? njt : 1;           // a whole object (Pw = the largest such prefix)
  const int Pr = P - Pw;                                            // pairs of the objects behind the whole rounds (0 < Pr < 2 G)
  int nb = Pr > 0 ? plan_div(G, Pr) : 1;                            // column blocks per pair that still fit one round
  if (nb > plan_div(njt, kSplitMinTiles)) nb = plan_div(njt, kSplitMinTiles);
  if (nb > kSplitMax - 4) nb = kSplitMax - 4;
  if (nb < 1) nb = 1;
  int c = plan_div(njt + nb - 1, nb);
  c = plan_div(c + cq - 1, cq) * cq;                                // (fp16 modes: a step is two tiles)
  return c > 0 ? c : 1;
}

// [r6] CHOICE OF THE PLAN when the launch's pairs fit its workgroups (P <= G).  Candidates, priced in tiles of one workgroup:
//   * the plain plan (aligned blocks of C tiles + remainder chunks that cross pair boundaries) at the smallest chunk length C0 whose
//     chunks fit the workgroups: the even cut;
//   * EQUALISED column blocks at the smallest chunk length c1 whose chunks fit (every object gets ceil(njt / c1) blocks of equal
//     length: no remainder chunks, every chunk one segment): taken when c1 is within 8 % (split: 4 %) of C0 -- the rule of rounds 3-5
//     (the bench launch: clips of 100-140 tiles at c1 = 122: the longest few cut in two, merged in place);
//   * WHOLE pairs (chunk length cw = the longest object: no partials, no merge) when its cost cw + half a
//     segment is within 1.15x (fp16 walks) / 1.0x (split) of the plain plan's C0 + a segment.  Measured (cold caches, one box,
//     tools/plan_ab.sh): 16 clips with boxes of 90-140 tiles, fp16: whole pairs (cw = 150) 137.7 us against 172.9 us for the plain
//     plan at C0 = 134 -- its multi-segment remainder chunks cost more than their model; 12 clips (C0 = 100): 131.8 against
//     115.2 us, the plain plan stays (12 clips with equal boxes of 120 tiles: 114.5 against 108.8 us at C0 = 94); split, 16 clips: 289.6
//     against 264.6 us at C0 = 118, the plain plan stays.
//   * NOT taken: equalised blocks in TWO rounds (a workgroup runs chunk c and chunk c + G): 12 clips 105.6 / 128 us warm / cold
//     against 97 / 115 us for the plain plan -- the second chunk's start-up and publish cost what the finer cut saves.
__host__ __device__ inline int bank_eq_chunk_len(int njt_max, int nb, int cq, int cmin) {
  int c = plan_div(njt_max + nb - 1, nb);
  c = plan_div(c + cq - 1, cq) * cq;
  return c > cmin ? c : cmin;
}
__host__ __device__ inline int bank_eq_count(int nqt, int njt, int c) { return njt > 0 ? nqt * plan_div(njt + c - 1, c) : 0; }
struct BankPlanPick { int blocks, C; };
// The equalised candidates of a launch, 64 of them (one per lane of bank.hip's planning wave): i < kBankEqByBlocks: the longest object
// in i + 1 column blocks; the others: chunk lengths from the even cut Clo upwards in 16 steps over a quarter of it (where the 8 % rule can still hold)
constexpr int kBankEqByBlocks = 48;
__host__ __device__ inline int bank_eq_candidate(int i, int njt_max, int Clo, int cq, int cmin) {
  if (i < kBankEqByBlocks) return bank_eq_chunk_len(njt_max, i + 1, cq, cmin);
  int st = plan_div(Clo + 63, 64);
  st = plan_div(st + cq - 1, cq) * cq;
  const int c = Clo + (i - kBankEqByBlocks) * st;
  return c > cmin ? c : cmin;
}
// c1: the smallest chunk length whose equalised chunks fit the workgroups (0: none); cw: the chunk length that leaves every pair whole
// (0: they do not fit); C0: the plain plan's (0: none)
__host__ __device__ inline BankPlanPick bank_plan_pick(int c1, int cw, int C0, int sc, int kTerms) {
  if (!C0) return BankPlanPick{(c1 || cw) ? 2 : 0, c1 ? c1 : cw};
  if (c1 && 25 * c1 <= (kTerms != 3 ? 27 : 26) * C0) return BankPlanPick{2, c1};
  if (cw && 40 * (cw + sc / 2) <= (kTerms != 3 ? 46 : 40) * (C0 + sc)) return BankPlanPick{2, cw};
  return BankPlanPick{0, C0};
}

// Smallest chunk length a launch may use: a chunk must amortise its prologue and 128 KB partial
// (kSplitMinTiles), and a pair may have at most kSplitMax partials (combine's LDS).
__host__ __device__ inline int bank_chunk_min(int njt_max) {
  const int cmin = (njt_max + kSplitMax - 4) / (kSplitMax - 3);   // njt / C <= kSplitMax - 3 aligned + <= 2 remainder
  return cmin > kSplitMinTiles ? cmin : kSplitMinTiles;
}

// Partial slots of a bank read.  Objects are planned in groups of kBankMaxObj per launch; a group of n
// objects needs at most kSplitTargetSlots chunks + n * nqt_max pair boundaries.
constexpr int kBankMaxObj = 64;
__host__ __device__ inline int bank_nqt_max(int hw) { return (hw + 63) / 64; }   // (the bank read has no mean slot: masked
                                                                                // query cells are filled from the slots' column sums)
__host__ __device__ inline int bank_group_slot0(int obj0, int hw) {
  return (obj0 / kBankMaxObj) * (kSplitTargetSlots + kBankMaxObj * bank_nqt_max(hw));
}
__host__ __device__ inline int bank_total_slots(int no, int hw) {
  const int rem = no % kBankMaxObj;
  return bank_group_slot0(no - rem, hw) + (rem ? kSplitTargetSlots + rem * bank_nqt_max(hw) : 0);
}

struct BankReadArgs {
  void* bank;
  int no, Tcap, h, w, T;
  const float *qk, *qv;
  const int32_t* qry_rects;
  float* out;
  float *ws_o, *ws_ml;
  int32_t* ws_plan;
  int slots;
  void* ws;
  size_t ws_bytes;
  hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_end = nullptr;
  int gate = 0;               // != 0: bk_main returns at once when the bank's overflow word is set
  const int32_t* T_dev = nullptr;   // optional device-resident frame counter added to T
  int f16 = 0;                // arithmetic: 0 split fp16 (three MFMA terms), 1 fp16 operands (hi planes only, one term),
                              // 2 fp16 operands with an exact query (RMNET_BANK_QX)
  int t0 = 0;                 // first slot read (chunked reads of a bank longer than one launch can take)
  float* ml_out = nullptr;    // optional [no][2][h*w]: soft-max state of the merged query cells (bank.hip: bk_chain)
};
int bank_max_frames_per_launch();
int launch_bank_ctl_clear(int32_t* words, int nwords, hipStream_t st);   // (a kernel: see bank.hip)
int bank_chain_max_chunks();
int launch_bank_ml_fill(float* ml, int no, int hw, float l0, hipStream_t st);
int launch_bank_chain(float* out, const float* tmp, const float* ml, int no, int hw, int nchunk, hipStream_t st);
size_t bank_read_ws_bytes_T(int no, int h, int w, int T);
int launch_bank_append(void* bank, int no, int Tcap, int h, int w, int slot, const float* k4,
                       const float* v4, const int32_t* rects, hipStream_t st, const int32_t* slot_dev = nullptr);
int launch_bank_stage(void* bank, int no, int Tcap, int h, int w, int slot0, int nf, const float* k4,
                      const float* v4, long long k_cs, long long k_os, long long v_cs, long long v_os,
                      const int32_t* rects, hipStream_t st, const int32_t* slot_dev = nullptr, bool colsum = true);   // nf frames of a strided [no,C,T,h,w] source
size_t bank_overflow_offset(int no, int Tcap, int h, int w);
int launch_bank_main(const BankReadArgs& a, hipStream_t st);   // bank.hip: the whole read (one launch per 64 objects)
size_t bank_read_ws_bytes(int no, int h, int w);
int launch_bank_read(BankReadArgs& a, hipStream_t st);         // memory_read.hip: workspace carving + launch_bank_main

inline int check_launch() { return hipGetLastError() == hipSuccess ? RMNET_OK : RMNET_E_LAUNCH; }

}  // namespace rmnet
