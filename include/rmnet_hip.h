/*
 * rmnet_hip.h -- C ABI of librmnet_hip.so: RMNet's per-frame hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  Every entry point replaces one native
 * interface (or one torch-op sequence) of the reference, hzxie/RMNet; paths below are relative
 * to the reference checkout.  Conventions for all calls:
 *   - plain pointers + sizes; all tensor pointers are DEVICE pointers unless the name ends in
 *     _host; tensors are dense row-major fp32 / int32 in the reference's own layouts;
 *   - the caller owns every buffer (inputs, outputs, workspace); nothing is allocated inside;
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *     no host synchronisation, so calls can be captured in a hipGraph;
 *   - return value: RMNET_OK (0) or a negative RMNET_E_* code -- errors are returned, never
 *     printed-and-ignored (the reference prints launch errors and carries on,
 *     extensions/reg_att_map_generator/reg_att_map_generator.cu:117-121);
 *   - re-entrant: no global mutable state, safe from one host thread per GPU
 *     (utils/eval_server.py:249-255 calls the op that way).
 */
#ifndef RMNET_HIP_H_
#define RMNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMNET_ABI_VERSION 6   /* 2: rmnet_bank_read_f32 takes a mutable bank; rmnet_bank_area_offset.  3: RMNET_MR_F16 / RMNET_BANK_F16,
                               * rmnet_bank_read_f32_at takes flags.  4: banks of any Tcap, chunked reads of T > 2048 (rmnet_bank_read_workspace_bytes_for);
                               * the read counts out-of-window query elements; sticky error bits + time-out word behind the overflow word.
                               * 5: RMNET_MR_QX / RMNET_BANK_QX (fp16 operands with an exact query); a pair whose merge timed out is
                               * written as NaN; the overflow word is a zero / non-zero flag, not an element count.
                               * 6: the bank keeps the largest affinity logit its reads have seen (third int32 of the control block); channels-last glue entries */

enum {
  RMNET_OK = 0,
  RMNET_E_INVALID_ARG = -1, /* null pointer, non-positive size, unsupported combination */
  RMNET_E_WORKSPACE = -2,   /* workspace pointer null or too small                        */
  RMNET_E_LAUNCH = -3,      /* hipGetLastError() after a launch / memcpy was not success   */
  RMNET_E_UNSUPPORTED = -4  /* shape outside what the kernels implement (see each call)    */
};

int rmnet_abi_version(void);
const char *rmnet_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * G1/G2  Regional attention map generator.
 * Replaces: pybind `reg_att_map_generator.forward(mask, prob_threshold, n_pts_threshold,
 *           n_bbox_loose_pixels)` -- extensions/reg_att_map_generator/reg_att_map_generator_cuda.cpp:26-38
 *           and its kernel, reg_att_map_generator.cu:15-123.
 *   mask     [B,K,H,W] f32 (soft masks; channel 0 = background, never inspected)
 *   att_map  [B,K,H,W] f32 out: 1 inside the loosened box of channel k>=1, else 0; channel 0 all 0.
 *            May be NULL: the full-resolution map is then not written (the fused regional read
 *            below only needs the boxes).
 *   bboxes   [B,K,4] int32 out: (x_min, x_max, y_min, y_max); channel 0 = (0,0,0,0).
 *   cell_rects [B,K,4] int32 out, may be NULL: the box expressed on the 1/`cell_stride` feature
 *            grid as (cx0, cx1, cy0, cy1) inclusive -- exactly the cells that survive
 *            F.interpolate(att_map padded by (pad_l, pad_t), scale_factor=1/cell_stride)
 *            (models/rmnet.py:245, 307, 356); empty boxes and channel 0 give (1,0,1,0).
 *            cells_h x cells_w is the feature-grid size.
 * Outputs are fully written (no pre-zeroing needed).  Integer outputs are bit-exact.
 * ------------------------------------------------------------------------------------------- */
size_t rmnet_region_map_workspace_bytes(int B, int K, int H, int W);
int rmnet_region_map_f32(const float *mask, int B, int K, int H, int W, float prob_threshold,
                         int n_pts_threshold, int n_bbox_loose_pixels, float *att_map,
                         int32_t *bboxes, int32_t *cell_rects, int pad_l, int pad_t,
                         int cell_stride, int cells_h, int cells_w, void *workspace,
                         size_t workspace_bytes, void *stream);

/* A1 + G1 fused: the boxes of the FLOW-WARPED previous mask,
 *   RMNet.get_att_map(prev_mask, flow) = att_map_generator(warp(prev_mask, flow))
 * (models/rmnet.py:252-287), without materialising the warped mask: flow [B,2,H,W] fp32 (x then y
 * displacement in pixels), everything else as rmnet_region_map_f32.  The warp follows the arithmetic
 * PyTorch-ROCm executes for warp(): bilinear, align_corners=True, zero padding, validity of the
 * sampling footprint thresholded at 0.9999.  `warped` (optional, [B,K,H,W], channel 0 untouched)
 * receives the warped mask itself.  Same workspace size as rmnet_region_map_f32. */
int rmnet_region_map_warped_f32(const float *mask, const float *flow, int B, int K, int H, int W,
                                float prob_threshold, int n_pts_threshold, int n_bbox_loose_pixels,
                                float *att_map, int32_t *bboxes, int32_t *cell_rects, int pad_l,
                                int pad_t, int cell_stride, int cells_h, int cells_w, float *warped,
                                void *workspace, size_t workspace_bytes, void *stream);

/* Box -> cell rectangle only (same formula as above), for boxes already on the device. */
int rmnet_boxes_to_cell_rects_i32(const int32_t *bboxes, int n_boxes, int k_per_batch, int pad_l,
                                  int pad_t, int cell_stride, int cells_h, int cells_w,
                                  int32_t *cell_rects, void *stream);

/* ---------------------------------------------------------------------------------------------
 * M1 (+M2/M3 fused)  Regional memory read.
 * Replaces: MemoryReader.forward, models/rmnet.py:147-165 (bmm, /sqrt(De), softmax over THW,
 *           bmm, cat) and -- when rectangles are given -- the K/V box masking around it,
 *           models/rmnet.py:244-248 (memory side) and :356-358 (query side).
 *   m_key  [no, De, T, h, w]   element (o,c,t,y,x) at o*m_obj_stride + c*m_chan_stride + t*h*w + y*w + x
 *   m_val  [no, Do, T, h, w]   same strides rule with v_obj_stride / v_chan_stride
 *          (contiguous NCTHW: chan_stride = T*h*w, obj_stride = C*T*h*w; a pre-allocated memory
 *           bank of capacity Tcap uses chan_stride = Tcap*h*w).  Pass 0 to mean "contiguous".
 *   q_key  [no, De, h, w], q_val [no, Do, h, w]   contiguous
 *   mem_val [no, 2*Do, h, w]  out: channels [0,Do) = read-out, [Do,2Do) = q_val (masked if regional)
 *   p_out  [no, T*h*w, h*w] out or NULL: the softmax affinity the reference returns as `viz`
 *          (models/rmnet.py:157,165); computed only when requested (extra pass).
 *   mem_rects [no, T, 4] int32 or NULL; qry_rects [no, 4] int32 or NULL: cell rectangles
 *          (cx0,cx1,cy0,cy1) inclusive.  When given, cells outside a rectangle are treated as
 *          if K, V (memory) / q_key, q_val (query) had been multiplied by 0 there -- the inputs
 *          may be the un-masked tensors (the multiply is fused) or already masked ones.
 *          Both NULL = dense read (drop-in MemoryReader).
 *   flags  RMNET_MR_* below.
 * The fast path needs De == 128 and Do == 512 (RMNet's sizes, models/rmnet.py:185-186); other
 * sizes run a generic (slow, still on-GPU) path that needs the p-sized workspace.
 * Fast path arithmetic: the memory is first staged (one launch, inside `workspace`) into a transient
 * split-fp16 bank (below) and read with 3-term fp16 MFMAs, fp32 accumulate -- fp32-class accuracy for
 * |K|, |V| < 1023.5.  The staging pass counts elements outside that window on the device; if there is
 * one (or a NaN / Inf), the exact-fp32 MFMA kernel (v_mfma_f32_16x16x4_f32, no range limit, ~4x slower)
 * does the read instead -- chosen on the device, no host synchronisation, no silent saturation.
 * RMNET_MR_EXACT_FP32 forces the exact kernel.  Parity with the reference is a tolerance (see tests).
 * RMNET_MR_F16 (opt-in) reads the staged bank with its hi planes only: K, V, the query and the soft-max weights enter
 * the MFMAs rounded to fp16 (11 significant bits), fp32 accumulate -- one MFMA term instead of three and half the bank
 * bytes, 1.5-2x as fast.  Error of a read-out: about 2^-11 of the values it averages (3e-4 relative at worst, a
 * peaked soft-max returning one cell's value; 5e-6 absolute when hundreds of cells contribute), against 1e-7 for the
 * default.  It meets the bar the reference's task sets (mask IoU within 1e-3 on whole clips, tests/test_gpu_parity.py);
 * it is NOT fp32-class.  Same range rules and the same device-side fallback as the default.
 * ------------------------------------------------------------------------------------------- */
#define RMNET_MR_DEFAULT 0
#define RMNET_MR_FORCE_GENERIC 1 /* use the generic path even for De=128, Do=512 (testing) */
#define RMNET_MR_EXACT_FP32 2    /* fast shape only: skip the split-fp16 bank, run the exact-fp32 MFMA kernel */
#define RMNET_MR_F16 4           /* fast shape only: fp16 operands (hi planes of the bank), fp32 accumulate */
#define RMNET_BANK_F16 4         /* the same switch for rmnet_bank_read_f32_at */
#define RMNET_MR_QX 8            /* fast shape only: RMNET_MR_F16 with an EXACT QUERY -- q enters the logits as a hi/lo pair (two MFMA terms), K, P
                                  * and V stay rounded to fp16.  q's rounding is the one logit error that is coherent over all memory cells of
                                  * a query; removing it recovers most of what RMNET_MR_F16 costs in mask IoU (measured on whole clips against
                                  * the CPU path: profiles/r05_iou_calibration.md) for 3 % (frame loop) to 9 % (back to back) of its speed.
                                  * Mutually exclusive with RMNET_MR_F16. */
#define RMNET_BANK_QX 8          /* the same switch for rmnet_bank_read_f32_at */

size_t rmnet_memory_read_workspace_bytes(int no, int De, int Do, int T, int h, int w, int flags);
int rmnet_memory_read_f32(const float *m_key, const float *m_val, const float *q_key,
                          const float *q_val, int no, int De, int Do, int T, int h, int w,
                          long long m_chan_stride, long long m_obj_stride, long long v_chan_stride,
                          long long v_obj_stride, float *mem_val, float *p_out,
                          const int32_t *mem_rects, const int32_t *qry_rects, int flags,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Same call, with optional HIP events recorded on `stream` around the two kernels of the fast
 * path: ev_start at the very start of the call (before the staging pass), ev_mid after the read kernel, ev_end after the combine kernel of the exact-fp32
 * path (which returns at once when the split-fp16 bank kernel did the whole read)
 * (each a hipEvent_t as void*, any may be NULL).  bench.py uses this to time the dominant kernel
 * on the stream it actually runs on; it changes nothing else. */
int rmnet_memory_read_f32_ev(const float *m_key, const float *m_val, const float *q_key,
                             const float *q_val, int no, int De, int Do, int T, int h, int w,
                             long long m_chan_stride, long long m_obj_stride,
                             long long v_chan_stride, long long v_obj_stride, float *mem_val,
                             float *p_out, const int32_t *mem_rects, const int32_t *qry_rects,
                             int flags, void *workspace, size_t workspace_bytes, void *stream,
                             void *ev_start, void *ev_mid, void *ev_end);

/* ---------------------------------------------------------------------------------------------
 * P1/P2 + M1-M3  Regional memory BANK: the form the frame loop uses.
 * Replaces: the K-slot padding, box masking and per-frame torch.cat of the whole memory
 *           (models/rmnet.py:191-205, 239-248, 416-426) plus the read itself (:361).
 * A bank is an opaque device buffer of rmnet_bank_bytes(no, Tcap, h, w) bytes holding, per object
 * and per memorised frame ("slot"), only the cells inside that frame's box, already split into
 * fp16 hi/lo planes in the MFMA fragment order, plus the slot's per-channel value sums (see
 * csrc/bank.hip).  De = 128, Do = 512.  THE CALLER ZERO-FILLS A NEW BANK (it carries an overflow word and
 * the read kernel's arrival tickets, which every read leaves at zero again).
 *   rmnet_bank_append_f32: write frame `slot` from k4 [no,128,h,w] / v4 [no,512,h,w] (fp32, the
 *       KeyValue outputs, UN-masked) and its cell rectangles rects [no,4] (NULL = whole frame).
 *       A slot may be overwritten (the tentative "previous frame" slot of models/rmnet.py:416-426).
 *   rmnet_bank_read_f32: read the first T slots with q_key [no,128,h,w], q_val [no,512,h,w] and
 *       query rectangles qry_rects [no,4] (NULL = all cells) -> mem_val [no,1024,h,w], identical in
 *       meaning to rmnet_memory_read_f32 with the same rectangles.  Arithmetic: split-fp16 MFMA
 *       (hi*hi + hi*lo + lo*hi), fp32 accumulate -- fp32-class accuracy.  ONE kernel launch per 64 objects
 *       does the whole of MemoryReader.forward (soft-max read, merge of the partial results by the last
 *       workgroup of each query tile, the read-out of masked query cells, the q_val half of the cat).
 *       The bank is not const: the launch uses its ticket words -- one stream at a time may read a given bank.
 *       ev_* as in rmnet_memory_read_f32_ev (may be NULL; ev_mid and ev_end now bracket nothing).
 * Range: K and V are stored times 2^6; elements with |x| >= 1023.5 saturate and NaN / Inf are lost.  Every
 *       16-byte group written with such an element increments the int32 overflow word that lives at byte
 *       rmnet_bank_overflow_offset() of the bank; a caller that cannot rule such inputs out checks it (once per
 *       clip is enough) and re-runs with rmnet_memory_read_f32(..., RMNET_MR_EXACT_FP32).
 *       The QUERY is split the same way after scaling by log2(e)/sqrt(128) * 2^6 (about 8.2): a q_key element inside the query box
 *       with |x| beyond ~8e3 (or NaN / Inf) is counted in the same overflow word by the read itself (ABI v4; it used to saturate
 *       silently), so "overflow word == 0 after the read" covers both sides.  The word also carries two sticky error bits:
 *       1 << 30 = an append / read through a device counter hit a slot or frame count outside [0, Tcap] (nothing was written /
 *       the count was clamped), 1 << 29 = a merge inside a read gave up waiting for another workgroup's partial (cannot happen on
 *       a healthy device; the int32 right behind the overflow word counts these time-outs separately; the read-out of the pair whose
 *       merge gave up is written as NaN, never as a partially merged value).  Any non-zero value means: do not trust the reads of
 *       this bank, re-run exactly.  The word is a FLAG (zero / non-zero), not a count of elements: the read adds to it once per
 *       (query element, segment of the launch plan) that sees it, the append once per 16-byte group.
 *       Logit word (ABI v6): the THIRD int32 of the control block (byte rmnet_bank_overflow_offset() + 8) holds, as float bits, the
 *       largest soft-max reference max_j S_ij * log2(e) any read of this bank has used so far (sticky maximum over reads, queries and
 *       objects; >= 0; the kernel's reference is deferred: the true maximum logit lies within 8 above value * ln 2).  The error of
 *       RMNET_BANK_F16 / RMNET_BANK_QX grows with the logits (K and q are rounded to 11 bits: ~2^-11 |S| per logit), so a caller that
 *       wants the fp32-class result on peaked soft-maxes checks this word once per clip and re-reads with flags = 0 when it is large
 *       (rmnet_amd/rmnet.py does, bound and evidence: profiles/r06_iou_temperature.md).  Zero in a new bank.
 *       rmnet_bank_area_offset(): byte offset of the int32 [no][Tcap] table of cells stored per slot (accounting).
 *       One launch reads at most 2048 slots (LDS prefix arrays).  Longer memories (models/rmnet.py:416-426 has no bound; ABI v4):
 *       a bank may have any Tcap; rmnet_bank_read_f32 / _at with T > 2048 (host-side T only: T_dev must be NULL for such a bank)
 *       read it in chunks of 2048 slots and merge the chunks' read-outs by their soft-max state (m, l) -- the same merge the
 *       kernel applies to the partial results of one launch -- in a workspace of rmnet_bank_read_workspace_bytes_for(...) bytes.
 * rmnet_bank_read_f32_at(..., flags): 0 = the arithmetic above; RMNET_BANK_F16 = hi planes only (see RMNET_MR_F16: fp16
 *       operands, fp32 accumulate, ~2^-11 relative, 1.5-2x as fast); RMNET_BANK_QX = the same with an exact query (see
 *       RMNET_MR_QX).  The bank is the same in every mode: a clip can be memorised once and read in all three.
 * ------------------------------------------------------------------------------------------- */
size_t rmnet_bank_bytes(int no, int Tcap, int h, int w);
size_t rmnet_bank_overflow_offset(int no, int Tcap, int h, int w);
size_t rmnet_bank_area_offset(int no, int Tcap, int h, int w);
int rmnet_bank_append_f32(void *bank, int no, int Tcap, int h, int w, int slot, const float *k4,
                          const float *v4, const int32_t *rects, void *stream);
/* The same two calls with a DEVICE-RESIDENT frame counter (int32, may be NULL): the slot written is slot + *slot_dev,
 * the frames read are T + *T_dev (clamped to [1, Tcap]).  The host need not know how long the memory is, so one
 * captured HIP graph of the frame loop (models/rmnet.py:410-450: memorise frame t-1 into slot `committed`, read
 * committed + 1 frames) can be replayed for every frame while the counter is bumped by a one-element add between
 * replays. */
int rmnet_bank_append_f32_at(void *bank, int no, int Tcap, int h, int w, int slot, const int32_t *slot_dev,
                             const float *k4, const float *v4, const int32_t *rects, void *stream);
int rmnet_bank_read_f32_at(void *bank, int no, int Tcap, int h, int w, int T, const int32_t *T_dev, int flags,
                           const float *q_key, const float *q_val, const int32_t *qry_rects,
                           float *mem_val, void *workspace, size_t workspace_bytes, void *stream,
                           void *ev_start, void *ev_mid, void *ev_end);
/* workspace of a read: the first form covers T <= 2048 frames, the _for form any T (ABI v4, chunked reads) */
size_t rmnet_bank_read_workspace_bytes(int no, int h, int w);
size_t rmnet_bank_read_workspace_bytes_for(int no, int h, int w, int T);
int rmnet_bank_read_f32(void *bank, int no, int Tcap, int h, int w, int T,
                        const float *q_key, const float *q_val, const int32_t *qry_rects,
                        float *mem_val, void *workspace, size_t workspace_bytes, void *stream,
                        void *ev_start, void *ev_mid, void *ev_end);

/* M2/M3 standalone: y = x * rectangle-mask, x [n, C, T, h, w] contiguous, rects [n, T, 4].
 * Replaces the elementwise multiplies at models/rmnet.py:247-248 and :357-358 when a caller
 * wants the masked tensors themselves (e.g. RMNet.memorize's return values). */
int rmnet_rect_mask_f32(const float *x, int n, int C, int T, int h, int w, const int32_t *rects,
                        float *y, void *stream);

/* C1 glue: per-channel affine + residual + ReLU in one pass over an NCHW fp32 activation,
 *   out[n,c,:] = act(x[n,c,:] * scale[c] + shift[c] + (res[n,c,:] * res_scale[c] + res_shift[c]))
 * scale / shift / res / res_scale / res_shift may each be NULL (1, 0, no residual, 1, 0); act = relu:
 * 0 none, 1 ReLU (NaN propagates like torch.relu), 2 LeakyReLU(0.1) (the slope of every activation of
 * models/tiny_flownet.py).  In place (out == x or out == res) is allowed.
 * Replaces the elementwise passes the reference runs after every convolution in eval mode:
 * BatchNorm2d + ReLU + skip add of the torchvision Bottleneck (models/rmnet.py:66-80, 96-103 use
 * resnet50's layers) and conv bias + ReLU + skip add of ResBlock (models/rmnet.py:24-48), with
 * scale/shift = gamma/sqrt(var+eps), beta - mean*scale  or  (NULL, conv bias). */
int rmnet_channel_affine_f32(const float *x, const float *scale, const float *shift,
                             const float *res, const float *res_scale, const float *res_shift,
                             int relu, long long N, int C, long long HW, float *out, void *stream);

/* C1 glue: out = max_pool2d(relu(x * scale[c] + shift[c]), kernel 3, stride 2, padding 1), x [N,C,H,W] ->
 * out [N,C,(H-1)/2+1,(W-1)/2+1] fp32 NCHW; scale / shift may be NULL.  Replaces bn1 -> relu -> maxpool of
 * the ResNet-50 stems (torchvision layers used at models/rmnet.py:74-76, 98-100) without writing the
 * full-resolution activation. */
int rmnet_affine_relu_maxpool_f32(const float *x, const float *scale, const float *shift, long long N,
                                  int C, int H, int W, float *out, void *stream);

/* C1 glue: out = skip + bilinear_x2(x), x [N,C,h,w] -> out / skip [N,C,2h,2w] fp32 NCHW, with
 * torch's align_corners=False source-index rule.  skip may be NULL (plain upsample); out may be skip.
 * Replaces F.interpolate(pm, scale_factor=2, mode='bilinear') and the add of Refine.forward
 * (models/rmnet.py:117-119). */
int rmnet_upsample2x_add_f32(const float *x, const float *skip, long long N, int C, int h, int w,
                             float *out, void *stream);

/* C1 glue, CHANNELS-LAST variants (ABI v6): the same three passes for activations that are [N, H, W, C] in memory -- what MIOpen's NHWC
 * convolution kernels produce and consume without layout transposes (torch: tensor.to(memory_format=torch.channels_last)).  Same arithmetic and
 * rounding as the NCHW entries above; C % 4 == 0, every pointer 16-byte aligned (RMNET_E_INVALID_ARG otherwise).
 *   rmnet_channel_affine_nhwc_f32:      x / res / out [rows = N*H*W][C]
 *   rmnet_upsample2x_add_nhwc_f32:      x [N,h,w,C] -> out / skip [N,2h,2w,C]
 *   rmnet_affine_relu_maxpool_nhwc_f32: x [N,H,W,C] -> out [N,(H-1)/2+1,(W-1)/2+1,C] */
int rmnet_channel_affine_nhwc_f32(const float *x, const float *scale, const float *shift, const float *res, const float *res_scale,
                                  const float *res_shift, int relu, long long rows, int C, float *out, void *stream);
int rmnet_upsample2x_add_nhwc_f32(const float *x, const float *skip, long long N, int C, int h, int w, float *out, void *stream);
int rmnet_affine_relu_maxpool_nhwc_f32(const float *x, const float *scale, const float *shift, long long N, int C, int H, int W,
                                       float *out, void *stream);

/* P3/P4 tail: decoder logits -> foreground probability -> soft aggregation -> un-pad (-> soft-max over
 * the K mask channels) in one pass.  dec [n_tot,2,Hp,Wp]: 2-class logits of the objects in flight;
 * clip b owns objects [obj_begin[b], obj_begin[b+1]) (device int32 [B+1]); logit / prob [B,K,H,W] with
 * H,W the un-padded size and (pad_l, pad_t) the padding removed on the left / top.  prob may be NULL.
 * Replaces F.softmax(logit, dim=1)[:, 1], RMNet.soft_aggregation, the un-pad slicing and the per-frame
 * F.softmax of the frame loop: models/rmnet.py:368-380, 289-302, 450. */
int rmnet_soft_aggregate_f32(const float *dec, const int32_t *obj_begin, int B, int K, int Hp, int Wp,
                             int pad_l, int pad_t, int H, int W, float *logit, float *prob,
                             void *stream);

/* ---------------------------------------------------------------------------------------------
 * F1  Optical-flow update after two affine warps.
 * Replaces: CPython `flow_affine_transformation.update_optical_flow(flow, M1, M2)` --
 *           extensions/flow_affine_transformation/flow_affine_transformation.cpp:39-90.
 *   flow [H,W,2] f32, m1/m2 [2,3] f32 -> out [H,W,2] f32 (integer-valued).  Bit-exact with the
 *   reference's fp32 arithmetic (no FMA contraction, round-half-away-from-zero), including its
 *   quirk that y1 is computed from the already-updated x1 (.cpp:72-73).
 * _host variant: all four pointers are HOST pointers (the reference's NumPy calling convention);
 * it stages through `workspace` (device, >= rmnet_flow_affine_workspace_bytes) and synchronises
 * the stream before returning.
 * ------------------------------------------------------------------------------------------- */
int rmnet_flow_affine_f32(const float *flow, const float *m1, const float *m2, int H, int W,
                          float *out, void *stream);
size_t rmnet_flow_affine_workspace_bytes(int H, int W);
int rmnet_flow_affine_f32_host(const float *flow_host, const float *m1_host, const float *m2_host,
                               int H, int W, float *out_host, void *workspace,
                               size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RMNET_HIP_H_ */
