// A client of librmnet_hip.so that is NOT Python: plain C++ + the HIP runtime, linking the C ABI of
// include/rmnet_hip.h exactly as a maintainer of the reference's C++ extensions would
// (extensions/reg_att_map_generator/reg_att_map_generator_cuda.cpp:26-38 calls one function per op).
// Build + run: tests/test_gpu_parity.py::test_c_abi_from_a_native_client
//   hipcc -O2 -std=c++17 tests/native/capi_client.cpp -Iinclude -Lrmnet_amd -lrmnet_hip -Wl,-rpath,$PWD/rmnet_amd
// Checks each op against a few lines of scalar host code (independent of oracle/): region map boxes + map,
// flow update, and the memory read (drop-in entry, dense and regional) on a small problem.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rmnet_hip.h"

#define HIP_OK(x) do { if ((x) != hipSuccess) { std::printf("HIP error line %d\n", __LINE__); return 2; } } while (0)
#define RM_OK(x) do { int rc_ = (x); if (rc_ != RMNET_OK) { std::printf("rmnet error %d (%s) line %d\n", rc_, rmnet_error_string(rc_), __LINE__); return 3; } } while (0)

template <class T> static T* dev_copy(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main() {
  if (rmnet_abi_version() != RMNET_ABI_VERSION) { std::printf("ABI mismatch\n"); return 1; }
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  unsigned seed = 7;

  // ---------------- G1: region map, B=1 K=3 40x70, two blobs
  {
    const int B = 1, K = 3, H = 40, W = 70;
    std::vector<float> mask((size_t)B * K * H * W, 0.0f);
    for (int y = 10; y < 25; ++y) for (int x = 30; x < 50; ++x) mask[(1 * H + y) * W + x] = 0.8f;
    for (int y = 2; y < 6; ++y) for (int x = 3; x < 9; ++x) mask[(2 * H + y) * W + x] = 0.5f;   // exactly at the threshold
    float* d_mask = dev_copy(mask);
    float* d_att; int32_t* d_bb; void* d_ws;
    HIP_OK(hipMalloc(&d_att, mask.size() * 4)); HIP_OK(hipMalloc(&d_bb, B * K * 4 * 4));
    const size_t wsb = rmnet_region_map_workspace_bytes(B, K, H, W);
    HIP_OK(hipMalloc(&d_ws, wsb));
    RM_OK(rmnet_region_map_f32(d_mask, B, K, H, W, 0.5f, 10, 4, d_att, d_bb, nullptr, 0, 0, 16, 1, 1, d_ws, wsb, st));
    std::vector<int32_t> bb(B * K * 4); std::vector<float> att(mask.size());
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(bb.data(), d_bb, bb.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(att.data(), d_att, att.size() * 4, hipMemcpyDeviceToHost));
    const int want[12] = {0, 0, 0, 0, 26, 53, 6, 28, 0, 12, 0, 9};     // (xmin, xmax, ymin, ymax), loosened by 4, clamped
    for (int i = 0; i < 12; ++i) if (bb[i] != want[i]) { std::printf("region map box %d: %d != %d\n", i, bb[i], want[i]); return 10; }
    for (int k = 0; k < K; ++k) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
      const float w = (k > 0 && x >= want[4 * k] && x <= want[4 * k + 1] && y >= want[4 * k + 2] && y <= want[4 * k + 3]) ? 1.0f : 0.0f;
      if (att[(k * H + y) * W + x] != w) { std::printf("region map att mismatch\n"); return 11; }
    }
    // the pybind layer's CHECKs become error codes here: a null mask is an invalid argument, not a crash
    if (rmnet_region_map_f32(nullptr, B, K, H, W, 0.5f, 10, 4, d_att, d_bb, nullptr, 0, 0, 16, 1, 1, d_ws, wsb, st) != RMNET_E_INVALID_ARG) return 12;
    if (rmnet_region_map_f32(d_mask, B, K, H, W, 0.5f, 10, 4, d_att, d_bb, nullptr, 0, 0, 16, 1, 1, d_ws, 1, st) != RMNET_E_WORKSPACE) return 13;
  }

  // ---------------- F1: flow update 20x30 (device entry and the host/NumPy-convention entry)
  {
    const int H = 20, W = 30;
    std::vector<float> flow((size_t)H * W * 2), m1 = {1.05f, 0.02f, 1.5f, -0.03f, 0.97f, -2.0f}, m2 = {0.98f, -0.01f, 0.5f, 0.02f, 1.01f, 1.25f};
    for (auto& v : flow) v = frand(seed) * 12.0f;
    std::vector<float> want(flow.size());
    auto rnd = [](float v) { return std::round(v); };                  // half away from zero, flow_affine_transformation.cpp:66-73
    auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
    for (int i = 0; i < H; ++i) for (int j = 0; j < W; ++j) {
      volatile float a;   // (keep every product and sum a separate fp32 rounding, as the reference's scalar loop does)
      float x2, y2, x1, y1;
      a = m2[0] * j; a = a + m2[1] * i; a = a + m2[2]; x2 = rnd(a);
      a = m2[3] * j; a = a + m2[4] * i; a = a + m2[5]; y2 = rnd(a);
      x1 = j + flow[(i * W + j) * 2]; y1 = i + flow[(i * W + j) * 2 + 1];
      a = m1[0] * x1; a = a + m1[1] * y1; a = a + m1[2]; const float nx1 = rnd(a);
      a = m1[3] * nx1; a = a + m1[4] * y1; a = a + m1[5]; const float ny1 = rnd(a);   // uses the UPDATED x1 (.cpp:72-73)
      const float cx1 = clampf(nx1, 0, W - 1), cy1 = clampf(ny1, 0, H - 1), cx2 = clampf(x2, 0, W - 1), cy2 = clampf(y2, 0, H - 1);
      want[(i * W + j) * 2] = cx1 - cx2; want[(i * W + j) * 2 + 1] = cy1 - cy2;
    }
    float *d_f = dev_copy(flow), *d_m1 = dev_copy(m1), *d_m2 = dev_copy(m2), *d_o; void* d_ws;
    HIP_OK(hipMalloc(&d_o, flow.size() * 4));
    RM_OK(rmnet_flow_affine_f32(d_f, d_m1, d_m2, H, W, d_o, st));
    std::vector<float> got(flow.size()), got2(flow.size());
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(got.data(), d_o, got.size() * 4, hipMemcpyDeviceToHost));
    const size_t wsb = rmnet_flow_affine_workspace_bytes(H, W);
    HIP_OK(hipMalloc(&d_ws, wsb));
    RM_OK(rmnet_flow_affine_f32_host(flow.data(), m1.data(), m2.data(), H, W, got2.data(), d_ws, wsb, st));
    for (size_t i = 0; i < want.size(); ++i)
      if (got[i] != want[i] || got2[i] != want[i]) { std::printf("flow mismatch at %zu: %g %g vs %g\n", i, got[i], got2[i], want[i]); return 20; }
  }

  // ---------------- M1 (+M2/M3): memory read through the drop-in entry, no=1 De=128 Do=512 T=2 4x6
  {
    const int no = 1, De = 128, Do = 512, T = 2, h = 4, w = 6, hw = h * w, thw = T * hw;
    std::vector<float> mk((size_t)De * thw), mv((size_t)Do * thw), qk((size_t)De * hw), qv((size_t)Do * hw);
    for (auto& v : mk) v = frand(seed); for (auto& v : mv) v = frand(seed) * 2; for (auto& v : qk) v = frand(seed); for (auto& v : qv) v = frand(seed);
    const int32_t mrect[8] = {1, 4, 0, 2, 0, 5, 1, 3}, qrect[4] = {0, 3, 1, 3};
    for (int regional = 0; regional < 2; ++regional) {
      // host reference: mask, affinity / sqrt(De), soft-max over the T*h*w memory cells, value read, cat with q_val
      auto keep_m = [&](int t, int c) { if (!regional) return 1.0f; const int y = c / w, x = c % w; const int32_t* r = mrect + 4 * t; return (x >= r[0] && x <= r[1] && y >= r[2] && y <= r[3]) ? 1.0f : 0.0f; };
      auto keep_q = [&](int c) { if (!regional) return 1.0f; const int y = c / w, x = c % w; return (x >= qrect[0] && x <= qrect[1] && y >= qrect[2] && y <= qrect[3]) ? 1.0f : 0.0f; };
      std::vector<double> want((size_t)2 * Do * hw);
      for (int i = 0; i < hw; ++i) {
        std::vector<double> s(thw);
        double mx = -1e300;
        for (int j = 0; j < thw; ++j) {
          double d = 0;
          for (int c = 0; c < De; ++c) d += (double)mk[(size_t)c * thw + j] * keep_m(j / hw, j % hw) * qk[(size_t)c * hw + i] * keep_q(i);
          s[j] = d / std::sqrt((double)De); mx = std::max(mx, s[j]);
        }
        double l = 0; for (int j = 0; j < thw; ++j) { s[j] = std::exp(s[j] - mx); l += s[j]; }
        for (int d = 0; d < Do; ++d) {
          double o = 0; for (int j = 0; j < thw; ++j) o += s[j] * mv[(size_t)d * thw + j] * keep_m(j / hw, j % hw);
          want[(size_t)d * hw + i] = o / l; want[(size_t)(Do + d) * hw + i] = qv[(size_t)d * hw + i] * keep_q(i);
        }
      }
      float *d_mk = dev_copy(mk), *d_mv = dev_copy(mv), *d_qk = dev_copy(qk), *d_qv = dev_copy(qv), *d_out; void* d_ws;
      std::vector<int32_t> mr(mrect, mrect + 8), qr(qrect, qrect + 4);
      int32_t *d_mr = dev_copy(mr), *d_qr = dev_copy(qr);
      HIP_OK(hipMalloc(&d_out, want.size() * 4));
      for (int flags : {RMNET_MR_DEFAULT, RMNET_MR_EXACT_FP32}) {
        const size_t wsb = rmnet_memory_read_workspace_bytes(no, De, Do, T, h, w, flags);
        HIP_OK(hipMalloc(&d_ws, wsb));
        RM_OK(rmnet_memory_read_f32(d_mk, d_mv, d_qk, d_qv, no, De, Do, T, h, w, 0, 0, 0, 0, d_out, nullptr, regional ? d_mr : nullptr,
                                    regional ? d_qr : nullptr, flags, d_ws, wsb, st));
        std::vector<float> got(want.size());
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t i = 0; i < want.size(); ++i) worst = std::max(worst, std::fabs((double)got[i] - want[i]));
        if (!(worst < 3e-5)) { std::printf("memory read (regional %d, flags %d): max err %g\n", regional, flags, worst); return 30; }
        HIP_OK(hipFree(d_ws));
      }
      // too small a workspace is reported, not overrun
      HIP_OK(hipMalloc(&d_ws, 256));
      if (rmnet_memory_read_f32(d_mk, d_mv, d_qk, d_qv, no, De, Do, T, h, w, 0, 0, 0, 0, d_out, nullptr, nullptr, nullptr, 0, d_ws, 256, st) != RMNET_E_WORKSPACE) return 31;
    }
  }
  std::printf("capi_client: all ops OK\n");
  return 0;
}
