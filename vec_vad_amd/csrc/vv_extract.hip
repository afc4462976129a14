// Cube extraction: crop + cv2.resize(INTER_LINEAR) of n boxes out of T decoded frames in one launch
// (reference vad_datasets.py:70-93 get_foreground; calc_optical_flow.py:46-59,82 whole-frame resizes).
// HBM-bound gather: every output element reads <= 4 source elements that neighbouring lanes share through L2/TCP.
// The uint8 path is bit-exact fixed point; the float path rounds every product and sum to fp32 (no FMA contraction, see
// the pragma below) like the C++ it replaces.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vecvad_hip.h"
#include "vv_common.h"

// Every product and sum below must be rounded separately, like the host arithmetic it replaces: forbid FMA contraction
// for this translation unit.  (ROCm's __fmul_rn/__fadd_rn are plain operators defined in a header, i.e. BEFORE this
// pragma, and do get fused after inlining -- hence ordinary operators here.)
#pragma clang fp contract(off)

namespace {

struct Tap {
  int s0, s1;
  float w0, w1;
};

// one axis of cv::resize's linear table: d -> (s0, s1, 1-f, f)
__device__ inline Tap axis_tap(int d, int dst, int src, bool horizontal) {
  double scale = 1.0 / ((double)dst / (double)src);
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f = f - (float)s;
  Tap t;
  if (horizontal) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    t.s0 = s;
    t.s1 = min(s + 1, src - 1);
  } else {
    t.s0 = min(max(s, 0), src - 1);
    t.s1 = min(max(s + 1, 0), src - 1);
  }
  t.w0 = 1.f - f;
  t.w1 = f;
  return t;
}

__device__ inline int fixed11(float w) {
  int v = __float2int_rn(w * 2048.f);
  return min(max(v, -32768), 32767);
}

template <typename T>
__global__ void __launch_bounds__(256) crop_resize_kernel(const T* __restrict__ frames, int nT, int H, int W, int C,
                                                          const int32_t* __restrict__ crops, int n, int oh, int ow,
                                                          T* __restrict__ out) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)n * nT * oh * ow;
  if (gid >= total) return;
  int dx = (int)(gid % ow);
  int dy = (int)((gid / ow) % oh);
  int t = (int)((gid / ((int64_t)ow * oh)) % nT);
  int i = (int)(gid / ((int64_t)ow * oh * nT));
  int x_min = crops[4 * i + 0], y_min = crops[4 * i + 1];
  int cw = crops[4 * i + 2] - x_min, ch = crops[4 * i + 3] - y_min;
  const T* src = frames + (((int64_t)t * H + y_min) * W + x_min) * C;
  int64_t rs = (int64_t)W * C;
  T* dst = out + gid * C;
  if (cw == ow && ch == oh) {                       // same size: plain copy
    for (int c = 0; c < C; ++c) dst[c] = src[dy * rs + (int64_t)dx * C + c];
    return;
  }
  if (cw == 2 * ow && ch == 2 * oh) {               // exact 2x decimation -> INTER_AREA
    const T* p = src + (2 * dy) * rs + (int64_t)(2 * dx) * C;
    for (int c = 0; c < C; ++c) {
      if constexpr (sizeof(T) == 1) {
        dst[c] = (T)(((int)p[c] + (int)p[C + c] + (int)p[rs + c] + (int)p[rs + C + c] + 2) >> 2);
      } else {
        dst[c] = (((p[c] + p[C + c]) + p[rs + c]) + p[rs + C + c]) * 0.25f;
      }
    }
    return;
  }
  Tap tx = axis_tap(dx, ow, cw, true), ty = axis_tap(dy, oh, ch, false);
  const T* r0 = src + ty.s0 * rs;
  const T* r1 = src + ty.s1 * rs;
  int64_t o0 = (int64_t)tx.s0 * C, o1 = (int64_t)tx.s1 * C;
  if constexpr (sizeof(T) == 1) {
    int a0 = fixed11(tx.w0), a1 = fixed11(tx.w1), b0 = fixed11(ty.w0), b1 = fixed11(ty.w1);
    for (int c = 0; c < C; ++c) {
      int h0 = (int)r0[o0 + c] * a0 + (int)r0[o1 + c] * a1;
      int h1 = (int)r1[o0 + c] * a0 + (int)r1[o1 + c] * a1;
      int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      dst[c] = (T)min(max(v, 0), 255);
    }
  } else {
    for (int c = 0; c < C; ++c) {
      float h0 = r0[o0 + c] * tx.w0 + r0[o1 + c] * tx.w1;
      float h1 = r1[o0 + c] * tx.w0 + r1[o1 + c] * tx.w1;
      dst[c] = h0 * ty.w0 + h1 * ty.w1;
    }
  }
}

}  // namespace

extern "C" int vv_crop_resize(const void* frames, int32_t is_f32, int32_t T, int32_t H, int32_t W, int32_t C,
                              const int32_t* crops, int32_t n, int32_t oh, int32_t ow, void* out, vv_stream stream) {
  if (!frames || !crops || !out || T <= 0 || H <= 0 || W <= 0 || C <= 0 || n < 0 || oh <= 0 || ow <= 0)
    return VV_ERR_BAD_ARG;
  if (n == 0) return VV_OK;
  int64_t total = (int64_t)n * T * oh * ow;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffff) return VV_ERR_BAD_ARG;
  if (is_f32) {
    VV_LAUNCH(crop_resize_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
              (const float*)frames, T, H, W, C, crops, n, oh, ow, (float*)out);
  } else {
    VV_LAUNCH(crop_resize_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
              (const uint8_t*)frames, T, H, W, C, crops, n, oh, ow, (uint8_t*)out);
  }
  VV_CHECK_LAUNCH();
  return VV_OK;
}
