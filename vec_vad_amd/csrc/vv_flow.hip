// FlowNet2's three native ops, forward only, written for gfx950 (NCHW fp32 contiguous, like the reference's cffi ABI).
//
//   vv_correlation_fwd  <- ops/correlation/src/correlation_cuda_kernel.cu:10-106 (+ channels_first repack :10-32)
//   vv_resample2d_fwd   <- ops/resample2d/src/Resample2d_kernel.cu:20-66
//   vv_channelnorm_fwd  <- ops/channelnorm/src/ChannelNorm_kernel.cu:19-51
//
// Correlation: the reference first materialises two zero-padded NHWC copies (2 x 16.5 MB at 1024x448) and then
// runs one 32-thread block per output pixel looping over 441 displacements.  Here one 256-thread workgroup owns
// one output row segment of 32 pixels: the first feature map's [C][32] slab stays in LDS for the whole block, the
// second map's displaced rows are streamed through LDS in 64-channel chunks, and padding is a bounds predicate
// (no temporaries).  Each thread accumulates 3 of the 21 x-displacements for one pixel over all channels.
#include "vv_common.h"

namespace {

constexpr int CORR_XT = 32;    // output pixels per block
constexpr int CORR_CC = 64;    // channels per streamed chunk of input2

__global__ void __launch_bounds__(VV_WG)
correlation_k1_kernel(const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out, const int C,
                      const int H, const int W, const int oC, const int oH, const int oW, const int pad, const int md,
                      const int s1, const int s2, const int dr, const int xtiles) {
  extern __shared__ float lds[];
  const int D = 2 * dr + 1;                 // displacements per axis
  const int SPAN = (CORR_XT - 1) * s1 + 2 * dr * s2 + 1;   // input2 columns touched by the tile
  const int SPANP = SPAN | 1;               // odd stride -> no bank conflicts
  float* a1 = lds;                          // [C][XT]
  float* a2 = lds + C * CORR_XT;            // [CC][SPANP]

  const int xt = blockIdx.x % xtiles;
  const int y = blockIdx.x / xtiles;
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int x0 = xt * CORR_XT;
  // un-padded input coordinates of the first map for output (y, x): y*s1 + md - pad  (kernel radius 0)
  const int yy1 = y * s1 + md - pad;
  const float* p1 = in1 + (int64_t)n * C * H * W;
  const float* p2 = in2 + (int64_t)n * C * H * W;

  for (int e = tid; e < C * CORR_XT; e += VV_WG) {
    const int ch = e / CORR_XT, xl = e % CORR_XT;
    const int xx = (x0 + xl) * s1 + md - pad;
    float v = 0.f;
    if (x0 + xl < oW && (unsigned)yy1 < (unsigned)H && (unsigned)xx < (unsigned)W) v = p1[((int64_t)ch * H + yy1) * W + xx];
    a1[e] = v;
  }

  const int xl = tid & 31, tig = tid >> 5;    // 8 displacement groups
  const float scale = 1.f / (float)C;        // nelems = kernel_size^2 * C  (correlation_cuda_kernel.cu:65)
  const int xbase = x0 * s1 + md - pad - dr * s2;   // un-padded column of a2[.][0]

  for (int tj = 0; tj < D; ++tj) {
    const int yy2 = yy1 + (tj - dr) * s2;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += CORR_CC) {
      __syncthreads();
      for (int e = tid; e < CORR_CC * SPAN; e += VV_WG) {
        const int ch = e / SPAN, xs = e % SPAN;
        const int xx = xbase + xs;
        float v = 0.f;
        if (c0 + ch < C && (unsigned)yy2 < (unsigned)H && (unsigned)xx < (unsigned)W)
          v = p2[((int64_t)(c0 + ch) * H + yy2) * W + xx];
        a2[ch * SPANP + xs] = v;
      }
      __syncthreads();
      const int cn = min(CORR_CC, C - c0);
      const float* q1 = a1 + c0 * CORR_XT + xl;
      const float* q2 = a2 + xl * s1;
#pragma unroll 4
      for (int ch = 0; ch < cn; ++ch) {
        const float v1 = q1[ch * CORR_XT];
        const float* r = q2 + ch * SPANP;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int ti = tig + 8 * k;
          if (ti < D) acc[k] = fmaf(v1, r[ti * s2], acc[k]);
        }
      }
    }
    if (x0 + xl < oW) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ti = tig + 8 * k;
        if (ti < D) out[(((int64_t)n * oC + tj * D + ti) * oH + y) * oW + x0 + xl] = acc[k] * scale;
      }
    }
  }
}

__global__ void __launch_bounds__(VV_WG)
resample2d_kernel(const int64_t npix, const float* __restrict__ img, const float* __restrict__ flow,
                  float* __restrict__ out, const int C, const int H, const int W, const int fH, const int fW) {
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= npix) return;
  const int x = (int)(e % fW);
  const int y = (int)((e / fW) % fH);
  const int64_t b = e / ((int64_t)fW * fH);
  const float dx = flow[((b * 2 + 0) * fH + y) * fW + x];
  const float dy = flow[((b * 2 + 1) * fH + y) * fW + x];
  const float xf = (float)x + dx, yf = (float)y + dy;
  const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
  // corners clamped with the OUTPUT extents, weights from the un-clamped coordinate (Resample2d_kernel.cu:45-53)
  const int xL = max(min((int)floorf(xf), fW - 1), 0);
  const int xR = max(min((int)(floorf(xf) + 1.f), fW - 1), 0);
  const int yT = max(min((int)floorf(yf), fH - 1), 0);
  const int yB = max(min((int)(floorf(yf) + 1.f), fH - 1), 0);
  // the reference mixes double literals (1. - alpha) with float data: products in double, float accumulator
  const double wTL = (1. - (double)alpha) * (1. - (double)beta), wTR = (double)alpha * (1. - (double)beta);
  const double wBL = (1. - (double)alpha) * (double)beta, wBR = (double)alpha * (double)beta;
  for (int c = 0; c < C; ++c) {
    const float* p = img + (b * C + c) * (int64_t)H * W;
    float val = 0.f;
    val = (float)((double)val + wTL * (double)p[(int64_t)yT * W + xL]);
    val = (float)((double)val + wTR * (double)p[(int64_t)yT * W + xR]);
    val = (float)((double)val + wBL * (double)p[(int64_t)yB * W + xL]);
    val = (float)((double)val + wBR * (double)p[(int64_t)yB * W + xR]);
    out[((b * C + c) * fH + y) * (int64_t)fW + x] = val;
  }
}

__global__ void __launch_bounds__(VV_WG)
channelnorm_kernel(const int64_t npix, const float* __restrict__ in, float* __restrict__ out, const int C,
                   const int64_t HW) {
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= npix) return;
  const int64_t b = e / HW, pix = e % HW;
  const float* p = in + b * C * HW + pix;
  float r = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = p[c * HW];
    r = fmaf(v, v, r);
  }
  out[e] = sqrtf(r);
}

}  // namespace

extern "C" int vv_correlation_out_shape(int32_t C, int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size,
                                        int32_t max_displacement, int32_t stride1, int32_t stride2, int32_t* oC,
                                        int32_t* oH, int32_t* oW) {
  (void)C;
  if (stride1 <= 0 || stride2 <= 0 || kernel_size <= 0) return VV_ERR_BAD_ARG;
  const int kr = (kernel_size - 1) / 2, br = kr + max_displacement;     // correlation_cuda.c:25-34
  const int d = (max_displacement / stride2) * 2 + 1;
  *oC = d * d;
  const int ph = H + 2 * pad_size - 2 * br, pw = W + 2 * pad_size - 2 * br;
  *oH = (ph + stride1 - 1) / stride1;
  *oW = (pw + stride1 - 1) / stride1;
  return (*oH > 0 && *oW > 0) ? VV_OK : VV_ERR_BAD_ARG;
}

extern "C" int vv_correlation_fwd(const float* in1, const float* in2, float* out, int32_t B, int32_t C, int32_t H,
                                  int32_t W, int32_t pad_size, int32_t kernel_size, int32_t max_displacement,
                                  int32_t stride1, int32_t stride2, int32_t corr_type_multiply, vv_stream stream) {
  if (!in1 || !in2 || !out) return VV_ERR_BAD_ARG;
  if (kernel_size != 1 || corr_type_multiply != 1) return VV_ERR_UNSUPPORTED;   // FlowNetC.py:24-30 uses exactly this
  int oC, oH, oW;
  int rc = vv_correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oC, &oH, &oW);
  if (rc) return rc;
  const int dr = max_displacement / stride2;
  if (2 * dr + 1 > 24) return VV_ERR_UNSUPPORTED;
  const int span = (CORR_XT - 1) * stride1 + 2 * dr * stride2 + 1;
  const size_t lds = ((size_t)C * CORR_XT + (size_t)CORR_CC * (span | 1)) * sizeof(float);
  if (lds > 160 * 1024) return VV_ERR_UNSUPPORTED;
  const int xtiles = (oW + CORR_XT - 1) / CORR_XT;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)correlation_k1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  VV_LAUNCH(correlation_k1_kernel, dim3(xtiles * oH, B), dim3(VV_WG), lds, (hipStream_t)stream, in1, in2, out, C,
                     H, W, oC, oH, oW, pad_size, max_displacement, stride1, stride2, dr, xtiles);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_resample2d_fwd(const float* img, const float* flow, float* out, int32_t B, int32_t C, int32_t H,
                                 int32_t W, int32_t fH, int32_t fW, int32_t kernel_size, vv_stream stream) {
  if (!img || !flow || !out) return VV_ERR_BAD_ARG;
  if (kernel_size != 1) return VV_ERR_UNSUPPORTED;                 // resample2d.py:8 default, flownet2.py:37-52
  if (fH > H || fW > W) return VV_ERR_BAD_ARG;
  const int64_t npix = (int64_t)B * fH * fW;
  VV_LAUNCH(resample2d_kernel, dim3((unsigned)((npix + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, (hipStream_t)stream,
                     npix, img, flow, out, C, H, W, fH, fW);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_channelnorm_fwd(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                                  int32_t norm_deg, vv_stream stream) {
  if (!in || !out) return VV_ERR_BAD_ARG;
  if (norm_deg != 2) return VV_ERR_UNSUPPORTED;                    // the kernel ignores norm_deg and always does L2
  const int64_t npix = (int64_t)B * H * W;
  VV_LAUNCH(channelnorm_kernel, dim3((unsigned)((npix + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, (hipStream_t)stream,
                     npix, in, out, C, (int64_t)H * W);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
