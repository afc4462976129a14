// FlowNet2 forward: the tensor plumbing between the five sub-networks as three NHWC kernels.
//
//   vv_flownet_prep      FlowNet2_src/models/flownet2.py:66-72   rgb_mean over (frames, H, W), (x - mean) / rgb_max, split of the
//                        two frames, torch.cat -> the 6-channel network input and the two 3-channel images FlowNetC's siamese
//                        stem reads, all NHWC, one pass over the input
//   vv_warp_pack12       flownet2.py:76-86 and 90-100: nn.Upsample(x4, bilinear) of the previous flow * div_flow, Resample2d of
//                        frame 2, ChannelNorm of the brightness error, torch.cat((x, warped, flow / div_flow, norm)) -> the
//                        12-channel input of FlowNetS, straight into the conv stack's NHWC buffer
//   vv_fusion_pack11     flownet2.py:105-136: the two nearest x4 up-samplings (x div_flow / : div_flow), two ChannelNorms of the
//                        flows, two warps + brightness errors, torch.cat of the 11 channels -> FlowNetFusion's input
//
// They replace ~45 ATen launches (sum / sub / div / cat / permute copies) plus the stand-alone resample2d / channelnorm /
// upsample4 launches and the NCHW<->NHWC round trips per forward.  All three are HBM-bound (<= 48 B written per pixel).
// The warp arithmetic is the one of vv_flow.hip's resample2d_kernel (Resample2d_kernel.cu:20-66, double products on float
// data), the norm is ChannelNorm_kernel.cu:19-51's (float accumulation, fmaf).
#include "vv_common.h"

namespace {

// nn.Upsample(scale_factor=4) of a 2-channel NHWC flow map [B,h,w,cs] evaluated at full-resolution pixel (y, x), times `scale`.
//   mode 0: nearest (flownet2.py:43-44)        mode 1: bilinear, align_corners=False (torch >= 0.4 default; the oracle's)
//   mode 2: bilinear, align_corners=True (what the authors' PyTorch 0.3 computed, README.md:10,64)
__device__ __forceinline__ float2 flow_up4(const float* __restrict__ f, const int cs, const int h, const int w, const int y,
                                           const int x, const int mode, const float scale) {
  if (mode == 0) {
    const float2 v = *reinterpret_cast<const float2*>(f + ((int64_t)min(y >> 2, h - 1) * w + min(x >> 2, w - 1)) * cs);
    return make_float2(v.x * scale, v.y * scale);
  }
  float sy, sx;
  if (mode == 1) {
    sy = fmaxf(((float)y + 0.5f) * 0.25f - 0.5f, 0.f);
    sx = fmaxf(((float)x + 0.5f) * 0.25f - 0.5f, 0.f);
  } else {   // align_corners: src = dst * (in - 1) / (out - 1)
    const float ry = h > 1 ? (float)(h - 1) / (float)(4 * h - 1) : 0.f, rx = w > 1 ? (float)(w - 1) / (float)(4 * w - 1) : 0.f;
    sy = ry * (float)y;
    sx = rx * (float)x;
  }
  const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
  const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float2 a = *reinterpret_cast<const float2*>(f + ((int64_t)y0 * w + x0) * cs);
  const float2 b = *reinterpret_cast<const float2*>(f + ((int64_t)y0 * w + x1) * cs);
  const float2 c = *reinterpret_cast<const float2*>(f + ((int64_t)y1 * w + x0) * cs);
  const float2 d = *reinterpret_cast<const float2*>(f + ((int64_t)y1 * w + x1) * cs);
  float2 r;
  r.x = (hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x)) * scale;
  r.y = (hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y)) * scale;
  return r;
}

// Resample2d (kernel_size 1) of a 3-channel NHWC image with pixel stride 4 at (x + dx, y + dy): Resample2d_kernel.cu:39-64
__device__ __forceinline__ float4 warp3(const float* __restrict__ img, const int H, const int W, const int y, const int x,
                                        const float dx, const float dy) {
  const float xf = (float)x + dx, yf = (float)y + dy;
  const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
  const int xL = max(min((int)floorf(xf), W - 1), 0);
  const int xR = max(min((int)(floorf(xf) + 1.f), W - 1), 0);
  const int yT = max(min((int)floorf(yf), H - 1), 0);
  const int yB = max(min((int)(floorf(yf) + 1.f), H - 1), 0);
  const double wTL = (1. - (double)alpha) * (1. - (double)beta), wTR = (double)alpha * (1. - (double)beta);
  const double wBL = (1. - (double)alpha) * (double)beta, wBR = (double)alpha * (double)beta;
  const float4 tl = *reinterpret_cast<const float4*>(img + ((int64_t)yT * W + xL) * 4);
  const float4 tr = *reinterpret_cast<const float4*>(img + ((int64_t)yT * W + xR) * 4);
  const float4 bl = *reinterpret_cast<const float4*>(img + ((int64_t)yB * W + xL) * 4);
  const float4 br = *reinterpret_cast<const float4*>(img + ((int64_t)yB * W + xR) * 4);
  float4 o;
#define VV_W1(c)                                           \
  {                                                        \
    float v = 0.f;                                         \
    v = (float)((double)v + wTL * (double)tl.c);           \
    v = (float)((double)v + wTR * (double)tr.c);           \
    v = (float)((double)v + wBL * (double)bl.c);           \
    v = (float)((double)v + wBR * (double)br.c);           \
    o.c = v;                                               \
  }
  VV_W1(x) VV_W1(y) VV_W1(z)
#undef VV_W1
  o.w = 0.f;
  return o;
}

__device__ __forceinline__ float norm3(const float a, const float b, const float c) {
  float r = 0.f;
  r = fmaf(a, a, r);
  r = fmaf(b, b, r);
  r = fmaf(c, c, r);
  return sqrtf(r);
}
__device__ __forceinline__ float norm2(const float a, const float b) {
  float r = 0.f;
  r = fmaf(a, a, r);
  r = fmaf(b, b, r);
  return sqrtf(r);
}

// ---- prep, pass 1: per-(image, colour, block) partial sums of the [B,3,2,H,W] input, float4 loads, fp64 partials
__global__ void __launch_bounds__(VV_WG)
prep_sum_kernel(const float* __restrict__ in, const int64_t n_per, const int nblk, double* __restrict__ part) {
  const int bc = blockIdx.y, blk = blockIdx.x;
  const float4* p = reinterpret_cast<const float4*>(in + (int64_t)bc * n_per);
  const int64_t n4 = n_per >> 2;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int64_t i = (int64_t)blk * VV_WG + threadIdx.x; i < n4; i += (int64_t)nblk * VV_WG) {
    const float4 v = p[i];
    s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
  }
  double s = ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
  for (int64_t i = (n4 << 2) + (int64_t)blk * VV_WG + threadIdx.x; i < n_per; i += (int64_t)nblk * VV_WG)
    s += (double)in[(int64_t)bc * n_per + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double red[VV_WG / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < VV_WG / 64; ++i) t += red[i];
    part[(int64_t)bc * nblk + blk] = t;
  }
}

// ---- prep, pass 2: (x - mean) / rgb_max, NCHW [B,3,2,H,W] -> x6 [B,H,W,8] (ch 0..2 frame 0, 3..5 frame 1, 6..7 zero),
//      img0 / img1 [B,H,W,4] (3 colours + zero)
__global__ void __launch_bounds__(VV_WG)
prep_apply_kernel(const float* __restrict__ in, const double* __restrict__ part, const int nblk, const int64_t HW,
                  const float rgb_max, float* __restrict__ x6, float* __restrict__ img0, float* __restrict__ img1) {
  const int b = blockIdx.y;
  __shared__ float mean[3];
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int i = 0; i < nblk; ++i) t += part[((int64_t)b * 3 + threadIdx.x) * nblk + i];
    mean[threadIdx.x] = (float)(t / (double)(2 * HW));
  }
  __syncthreads();
  const int64_t pix = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (pix >= HW) return;
  const float* p = in + (int64_t)b * 6 * HW + pix;
  float v[6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    v[c] = (p[(int64_t)(2 * c) * HW] - mean[c]) / rgb_max;          // frame 0
    v[3 + c] = (p[(int64_t)(2 * c + 1) * HW] - mean[c]) / rgb_max;  // frame 1
  }
  const int64_t o = (int64_t)b * HW + pix;
  reinterpret_cast<float4*>(x6)[o * 2] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(x6)[o * 2 + 1] = make_float4(v[4], v[5], 0.f, 0.f);
  reinterpret_cast<float4*>(img0)[o] = make_float4(v[0], v[1], v[2], 0.f);
  reinterpret_cast<float4*>(img1)[o] = make_float4(v[3], v[4], v[5], 0.f);
}

__global__ void __launch_bounds__(VV_WG)
warp_pack12_kernel(const float* __restrict__ x6, const float* __restrict__ img1, const float* __restrict__ flow2, const int fcs,
                   const int H, const int W, const int mode, const float scale, const float div_flow,
                   float* __restrict__ out) {
  const int b = blockIdx.y;
  const int64_t pix = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (pix >= (int64_t)H * W) return;
  const int y = (int)(pix / W), x = (int)(pix % W);
  const int h = H >> 2, w = W >> 2;
  const float2 fl = flow_up4(flow2 + (int64_t)b * h * w * fcs, fcs, h, w, y, x, mode, scale);
  const int64_t o = (int64_t)b * H * W + pix;
  const float4 a0 = reinterpret_cast<const float4*>(x6)[o * 2], a1 = reinterpret_cast<const float4*>(x6)[o * 2 + 1];
  const float4 wr = warp3(img1 + (int64_t)b * H * W * 4, H, W, y, x, fl.x, fl.y);
  const float nrm = norm3(a0.x - wr.x, a0.y - wr.y, a0.z - wr.z);
  float4* q = reinterpret_cast<float4*>(out) + o * 3;
  q[0] = a0;
  q[1] = make_float4(a1.x, a1.y, wr.x, wr.y);
  q[2] = make_float4(wr.z, fl.x / div_flow, fl.y / div_flow, nrm);
}

__global__ void __launch_bounds__(VV_WG)
fusion_pack11_kernel(const float* __restrict__ x6, const float* __restrict__ img1, const float* __restrict__ s2f,
                     const int s2cs, const float* __restrict__ sdf, const int sdcs, const int H, const int W,
                     const float div_flow, float* __restrict__ out) {
  // concat3 = (x1, sd_flow, s2_flow, norm_sd, norm_s2, diff_sd, diff_s2), flownet2.py:132-136; both flows nearest x4
  const int b = blockIdx.y;
  const int64_t pix = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (pix >= (int64_t)H * W) return;
  const int y = (int)(pix / W), x = (int)(pix % W);
  const int h = H >> 2, w = W >> 2;
  const float2 s2 = flow_up4(s2f + (int64_t)b * h * w * s2cs, s2cs, h, w, y, x, 0, div_flow);
  float2 sd = flow_up4(sdf + (int64_t)b * h * w * sdcs, sdcs, h, w, y, x, 0, 1.f);
  sd.x = sd.x / div_flow;                      // flownet2.py:122 divides
  sd.y = sd.y / div_flow;
  const int64_t o = (int64_t)b * H * W + pix;
  const float4 a0 = reinterpret_cast<const float4*>(x6)[o * 2];
  const float* im = img1 + (int64_t)b * H * W * 4;
  const float4 w2 = warp3(im, H, W, y, x, s2.x, s2.y);
  const float4 wd = warp3(im, H, W, y, x, sd.x, sd.y);
  float4* q = reinterpret_cast<float4*>(out) + o * 3;
  q[0] = make_float4(a0.x, a0.y, a0.z, sd.x);
  q[1] = make_float4(sd.y, s2.x, s2.y, norm2(sd.x, sd.y));
  q[2] = make_float4(norm2(s2.x, s2.y), norm3(a0.x - wd.x, a0.y - wd.y, a0.z - wd.z),
                     norm3(a0.x - w2.x, a0.y - w2.y, a0.z - w2.z), 0.f);
}

}  // namespace

extern "C" int64_t vv_flownet_prep_workspace_bytes(int32_t B) { return (int64_t)B * 3 * 64 * sizeof(double); }

extern "C" int vv_flownet_prep(const float* inputs, int32_t B, int32_t H, int32_t W, float rgb_max, void* workspace,
                               int64_t workspace_bytes, float* x6, float* img0, float* img1, vv_stream stream) {
  if (!inputs || !workspace || !x6 || !img0 || !img1 || B <= 0 || H <= 0 || W <= 0) return VV_ERR_BAD_ARG;
  if (workspace_bytes < vv_flownet_prep_workspace_bytes(B)) return VV_ERR_BAD_ARG;
  if (((uintptr_t)inputs | (uintptr_t)x6 | (uintptr_t)img0 | (uintptr_t)img1) & 15) return VV_ERR_BAD_ARG;
  const int nblk = 64;
  const int64_t HW = (int64_t)H * W;
  double* part = reinterpret_cast<double*>(workspace);
  VV_LAUNCH(prep_sum_kernel, dim3(nblk, B * 3), dim3(VV_WG), 0, (hipStream_t)stream, inputs, 2 * HW, nblk, part);
  VV_CHECK_LAUNCH();
  VV_LAUNCH(prep_apply_kernel, dim3((unsigned)((HW + VV_WG - 1) / VV_WG), B), dim3(VV_WG), 0, (hipStream_t)stream, inputs, part,
            nblk, HW, rgb_max, x6, img0, img1);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_warp_pack12(const float* x6, const float* img1, const float* flow2, int32_t flow_cstride, int32_t B, int32_t H,
                              int32_t W, int32_t mode, float scale, float div_flow, float* out12, vv_stream stream) {
  if (!x6 || !img1 || !flow2 || !out12 || B <= 0 || H % 4 || W % 4 || flow_cstride < 2 || flow_cstride % 2) return VV_ERR_BAD_ARG;
  if (mode < 0 || mode > 2) return VV_ERR_BAD_ARG;
  const int64_t HW = (int64_t)H * W;
  VV_LAUNCH(warp_pack12_kernel, dim3((unsigned)((HW + VV_WG - 1) / VV_WG), B), dim3(VV_WG), 0, (hipStream_t)stream, x6, img1,
            flow2, flow_cstride, H, W, mode, scale, div_flow, out12);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_fusion_pack11(const float* x6, const float* img1, const float* s2_flow2, int32_t s2_cstride,
                                const float* sd_flow2, int32_t sd_cstride, int32_t B, int32_t H, int32_t W, float div_flow,
                                float* out12, vv_stream stream) {
  if (!x6 || !img1 || !s2_flow2 || !sd_flow2 || !out12 || B <= 0 || H % 4 || W % 4) return VV_ERR_BAD_ARG;
  if (s2_cstride < 2 || s2_cstride % 2 || sd_cstride < 2 || sd_cstride % 2) return VV_ERR_BAD_ARG;
  const int64_t HW = (int64_t)H * W;
  VV_LAUNCH(fusion_pack11_kernel, dim3((unsigned)((HW + VV_WG - 1) / VV_WG), B), dim3(VV_WG), 0, (hipStream_t)stream, x6, img1,
            s2_flow2, s2_cstride, sd_flow2, sd_cstride, H, W, div_flow, out12);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
