// capi.hip -- the extern "C" boundary declared in include/rmnet_hip.h.
#include "common.h"

using namespace rmnet;

extern "C" {

int rmnet_abi_version(void) { return RMNET_ABI_VERSION; }

const char* rmnet_error_string(int code) {
  switch (code) {
    case RMNET_OK: return "ok";
    case RMNET_E_INVALID_ARG: return "invalid argument (null pointer, non-positive size or inconsistent options)";
    case RMNET_E_WORKSPACE: return "workspace missing or too small";
    case RMNET_E_LAUNCH: return "HIP launch or copy failed";
    case RMNET_E_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    default: return "unknown error code";
  }
}

size_t rmnet_region_map_workspace_bytes(int B, int K, int H, int W) {
  if (B <= 0 || K <= 0 || H <= 0 || W <= 0) return 0;
  return region_map_ws_bytes(B, K, H, W);
}

int rmnet_region_map_f32(const float* mask, int B, int K, int H, int W, float prob_threshold,
                         int n_pts_threshold, int n_bbox_loose_pixels, float* att_map,
                         int32_t* bboxes, int32_t* cell_rects, int pad_l, int pad_t,
                         int cell_stride, int cells_h, int cells_w, void* workspace,
                         size_t workspace_bytes, void* stream) {
  return launch_region_map(mask, B, K, H, W, prob_threshold, n_pts_threshold, n_bbox_loose_pixels,
                           att_map, bboxes, cell_rects, pad_l, pad_t, cell_stride, cells_h, cells_w,
                           workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int rmnet_boxes_to_cell_rects_i32(const int32_t* bboxes, int n_boxes, int k_per_batch, int pad_l,
                                  int pad_t, int cell_stride, int cells_h, int cells_w,
                                  int32_t* cell_rects, void* stream) {
  return launch_boxes_to_rects(bboxes, n_boxes, k_per_batch, pad_l, pad_t, cell_stride, cells_h,
                               cells_w, cell_rects, static_cast<hipStream_t>(stream));
}

size_t rmnet_memory_read_workspace_bytes(int no, int De, int Do, int T, int h, int w, int flags) {
  if (no <= 0 || De <= 0 || Do <= 0 || T <= 0 || h <= 0 || w <= 0) return 0;
  return memory_read_ws_bytes(no, De, Do, T, h, w, flags);
}

int rmnet_memory_read_f32_ev(const float* m_key, const float* m_val, const float* q_key,
                             const float* q_val, int no, int De, int Do, int T, int h, int w,
                             long long m_chan_stride, long long m_obj_stride,
                             long long v_chan_stride, long long v_obj_stride, float* mem_val,
                             float* p_out, const int32_t* mem_rects, const int32_t* qry_rects,
                             int flags, void* workspace, size_t workspace_bytes, void* stream,
                             void* ev_start, void* ev_mid, void* ev_end) {
  if ((flags & RMNET_MR_F16) && (flags & RMNET_MR_QX)) return RMNET_E_INVALID_ARG;   // one arithmetic per call
  MemReadArgs a;
  a.mk = m_key; a.mv = m_val; a.qk = q_key; a.qv = q_val;
  a.out = mem_val; a.p_out = p_out;
  a.mem_rects = mem_rects; a.qry_rects = qry_rects;
  a.no = no; a.De = De; a.Do = Do; a.T = T; a.h = h; a.w = w;
  a.mk_cs = m_chan_stride; a.mk_os = m_obj_stride; a.mv_cs = v_chan_stride; a.mv_os = v_obj_stride;
  a.flags = flags; a.ws = workspace; a.ws_bytes = workspace_bytes;
  a.ev_start = static_cast<hipEvent_t>(ev_start);
  a.ev_mid = static_cast<hipEvent_t>(ev_mid);
  a.ev_end = static_cast<hipEvent_t>(ev_end);
  return launch_memory_read(a, static_cast<hipStream_t>(stream));
}

int rmnet_memory_read_f32(const float* m_key, const float* m_val, const float* q_key,
                          const float* q_val, int no, int De, int Do, int T, int h, int w,
                          long long m_chan_stride, long long m_obj_stride, long long v_chan_stride,
                          long long v_obj_stride, float* mem_val, float* p_out,
                          const int32_t* mem_rects, const int32_t* qry_rects, int flags,
                          void* workspace, size_t workspace_bytes, void* stream) {
  return rmnet_memory_read_f32_ev(m_key, m_val, q_key, q_val, no, De, Do, T, h, w, m_chan_stride,
                                  m_obj_stride, v_chan_stride, v_obj_stride, mem_val, p_out,
                                  mem_rects, qry_rects, flags, workspace, workspace_bytes, stream,
                                  nullptr, nullptr, nullptr);
}

size_t rmnet_bank_bytes(int no, int Tcap, int h, int w) {
  if (no <= 0 || Tcap <= 0 || h <= 0 || w <= 0) return 0;
  return bank_bytes(no, Tcap, h, w);
}

size_t rmnet_bank_area_offset(int no, int Tcap, int h, int w) {
  if (no <= 0 || Tcap <= 0 || h <= 0 || w <= 0) return 0;
  return bank_area_offset(no, Tcap, h, w);
}

size_t rmnet_bank_overflow_offset(int no, int Tcap, int h, int w) {
  if (no <= 0 || Tcap <= 0 || h <= 0 || w <= 0) return 0;
  return bank_overflow_offset(no, Tcap, h, w);
}

int rmnet_bank_append_f32(void* bank, int no, int Tcap, int h, int w, int slot, const float* k4,
                          const float* v4, const int32_t* rects, void* stream) {
  return launch_bank_append(bank, no, Tcap, h, w, slot, k4, v4, rects, static_cast<hipStream_t>(stream));
}

int rmnet_bank_append_f32_at(void* bank, int no, int Tcap, int h, int w, int slot, const int32_t* slot_dev,
                             const float* k4, const float* v4, const int32_t* rects, void* stream) {
  return launch_bank_append(bank, no, Tcap, h, w, slot, k4, v4, rects, static_cast<hipStream_t>(stream), slot_dev);
}

size_t rmnet_bank_read_workspace_bytes(int no, int h, int w) {
  if (no <= 0 || h <= 0 || w <= 0) return 0;
  return bank_read_ws_bytes(no, h, w);
}

size_t rmnet_bank_read_workspace_bytes_for(int no, int h, int w, int T) {
  if (no <= 0 || h <= 0 || w <= 0 || T <= 0) return 0;
  return bank_read_ws_bytes_T(no, h, w, T);
}

int rmnet_bank_read_f32(void* bank, int no, int Tcap, int h, int w, int T, const float* q_key,
                        const float* q_val, const int32_t* qry_rects, float* mem_val,
                        void* workspace, size_t workspace_bytes, void* stream, void* ev_start,
                        void* ev_mid, void* ev_end) {
  BankReadArgs a;
  a.bank = bank; a.no = no; a.Tcap = Tcap; a.h = h; a.w = w; a.T = T;
  a.qk = q_key; a.qv = q_val; a.qry_rects = qry_rects; a.out = mem_val;
  a.ws_o = nullptr; a.ws_ml = nullptr; a.ws_plan = nullptr; a.slots = 0;
  a.ws = workspace; a.ws_bytes = workspace_bytes;
  a.ev_start = static_cast<hipEvent_t>(ev_start);
  a.ev_mid = static_cast<hipEvent_t>(ev_mid);
  a.ev_end = static_cast<hipEvent_t>(ev_end);
  return launch_bank_read(a, static_cast<hipStream_t>(stream));
}

int rmnet_bank_read_f32_at(void* bank, int no, int Tcap, int h, int w, int T, const int32_t* T_dev, int flags,
                           const float* q_key, const float* q_val, const int32_t* qry_rects, float* mem_val,
                           void* workspace, size_t workspace_bytes, void* stream, void* ev_start, void* ev_mid,
                           void* ev_end) {
  BankReadArgs a;
  if (flags != 0 && flags != RMNET_BANK_F16 && flags != RMNET_BANK_QX) return RMNET_E_INVALID_ARG;
  a.bank = bank; a.no = no; a.Tcap = Tcap; a.h = h; a.w = w; a.T = T; a.T_dev = T_dev;
  a.f16 = (flags & RMNET_BANK_F16) ? 1 : (flags & RMNET_BANK_QX) ? 2 : 0;
  a.qk = q_key; a.qv = q_val; a.qry_rects = qry_rects; a.out = mem_val;
  a.ws_o = nullptr; a.ws_ml = nullptr; a.ws_plan = nullptr; a.slots = 0;
  a.ws = workspace; a.ws_bytes = workspace_bytes;
  a.ev_start = static_cast<hipEvent_t>(ev_start);
  a.ev_mid = static_cast<hipEvent_t>(ev_mid);
  a.ev_end = static_cast<hipEvent_t>(ev_end);
  return launch_bank_read(a, static_cast<hipStream_t>(stream));
}

int rmnet_rect_mask_f32(const float* x, int n, int C, int T, int h, int w, const int32_t* rects,
                        float* y, void* stream) {
  return launch_rect_mask(x, n, C, T, h, w, rects, y, static_cast<hipStream_t>(stream));
}

int rmnet_channel_affine_f32(const float* x, const float* scale, const float* shift, const float* res,
                             const float* res_scale, const float* res_shift, int relu, long long N,
                             int C, long long HW, float* out, void* stream) {
  return launch_channel_affine(x, scale, shift, res, res_scale, res_shift, relu, N, C, HW, out,
                               static_cast<hipStream_t>(stream));
}

int rmnet_upsample2x_add_f32(const float* x, const float* skip, long long N, int C, int h, int w,
                             float* out, void* stream) {
  return launch_upsample2x_add(x, skip, N, C, h, w, out, static_cast<hipStream_t>(stream));
}

int rmnet_soft_aggregate_f32(const float* dec, const int32_t* obj_begin, int B, int K, int Hp, int Wp,
                             int pad_l, int pad_t, int H, int W, float* logit, float* prob,
                             void* stream) {
  return launch_soft_aggregate(dec, obj_begin, B, K, Hp, Wp, pad_l, pad_t, H, W, logit, prob,
                               static_cast<hipStream_t>(stream));
}

int rmnet_region_map_warped_f32(const float* mask, const float* flow, int B, int K, int H, int W,
                                float prob_threshold, int n_pts_threshold, int n_bbox_loose_pixels,
                                float* att_map, int32_t* bboxes, int32_t* cell_rects, int pad_l,
                                int pad_t, int cell_stride, int cells_h, int cells_w, float* warped,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!flow) return RMNET_E_INVALID_ARG;
  return launch_region_map_warped(mask, flow, B, K, H, W, prob_threshold, n_pts_threshold,
                                  n_bbox_loose_pixels, att_map, bboxes, cell_rects, pad_l, pad_t,
                                  cell_stride, cells_h, cells_w, warped, workspace, workspace_bytes,
                                  static_cast<hipStream_t>(stream));
}

int rmnet_affine_relu_maxpool_f32(const float* x, const float* scale, const float* shift, long long N,
                                  int C, int H, int W, float* out, void* stream) {
  return launch_affine_relu_maxpool(x, scale, shift, N, C, H, W, out, static_cast<hipStream_t>(stream));
}

int rmnet_channel_affine_nhwc_f32(const float* x, const float* scale, const float* shift, const float* res, const float* res_scale,
                                  const float* res_shift, int relu, long long rows, int C, float* out, void* stream) {
  return launch_channel_affine_nhwc(x, scale, shift, res, res_scale, res_shift, relu, rows, C, out, static_cast<hipStream_t>(stream));
}

int rmnet_upsample2x_add_nhwc_f32(const float* x, const float* skip, long long N, int C, int h, int w, float* out, void* stream) {
  return launch_upsample2x_add_nhwc(x, skip, N, C, h, w, out, static_cast<hipStream_t>(stream));
}

int rmnet_affine_relu_maxpool_nhwc_f32(const float* x, const float* scale, const float* shift, long long N, int C, int H, int W, float* out,
                                       void* stream) {
  return launch_affine_relu_maxpool_nhwc(x, scale, shift, N, C, H, W, out, static_cast<hipStream_t>(stream));
}

int rmnet_flow_affine_f32(const float* flow, const float* m1, const float* m2, int H, int W,
                          float* out, void* stream) {
  return launch_flow_affine(flow, m1, m2, H, W, out, static_cast<hipStream_t>(stream));
}

size_t rmnet_flow_affine_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  return (size_t)H * W * 2 * sizeof(float) * 2 + 256;
}

int rmnet_flow_affine_f32_host(const float* flow_host, const float* m1_host, const float* m2_host,
                               int H, int W, float* out_host, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!flow_host || !m1_host || !m2_host || !out_host || H <= 0 || W <= 0)
    return RMNET_E_INVALID_ARG;
  if (!workspace || workspace_bytes < rmnet_flow_affine_workspace_bytes(H, W))
    return RMNET_E_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t nb = (size_t)H * W * 2 * sizeof(float);
  char* base = static_cast<char*>(workspace);
  float* d_m = reinterpret_cast<float*>(base);            // 12 floats, padded to 256 B
  float* d_in = reinterpret_cast<float*>(base + 256);
  float* d_out = reinterpret_cast<float*>(base + 256 + nb);
  float mats[12];
  for (int i = 0; i < 6; ++i) { mats[i] = m1_host[i]; mats[6 + i] = m2_host[i]; }
  if (hipMemcpyAsync(d_m, mats, sizeof(mats), hipMemcpyHostToDevice, st) != hipSuccess) return RMNET_E_LAUNCH;
  if (hipMemcpyAsync(d_in, flow_host, nb, hipMemcpyHostToDevice, st) != hipSuccess) return RMNET_E_LAUNCH;
  if (int e = launch_flow_affine(d_in, d_m, d_m + 6, H, W, d_out, st)) return e;
  if (hipMemcpyAsync(out_host, d_out, nb, hipMemcpyDeviceToHost, st) != hipSuccess) return RMNET_E_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return RMNET_E_LAUNCH;
  return RMNET_OK;
}

}  // extern "C"
