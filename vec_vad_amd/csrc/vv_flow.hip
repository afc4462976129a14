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

// ---------------------------------------------------------------------------------------------------------------
// FlowNetC's correlation layer on NHWC maps (pad 20, kernel 1, max_displacement 20, stride1 1, stride2 2:
// FlowNetC.py:41-47), with the 1/C normalisation (correlation_cuda_kernel.cu:65,93-99) and FlowNetC's LeakyReLU
// (FlowNetC.py:92-93) fused, written straight into a channel slice of the consumer's NHWC concat buffer.
//
//   out[y][x][ti*21 + tj] = leaky( 1/C * sum_c f1[y][x][c] * f2[y + 2(ti-10)][x + 2(tj-10)][c] )
//
// fp32 VALU kernel (an MFMA Gram-matrix formulation would throw away 2/3 - 3/4 of every 16x16 / 32x32 block of the 21-wide band,
// and the fp32 matrix instruction runs at the vector rate anyway).  One workgroup = one output row y and DYB of the 21 row
// displacements, one row displacement per 64-lane wave (two per wave at W = 64); a thread owns two same-parity columns (x, x+2)
// and all 21 column displacements: 42 accumulators whose 22-wide window of f2 is shared by both columns -> 24 ds_read_b128 per
// 168 multiply-adds.  Both maps go through LDS in 16-channel chunks as float4 planes [c4][pos mod 4][pos div 4] (+1 float4 of
// plane padding): a wave's ds_read_b128 of "position 4j + const" walks consecutive float4 slots (conflict free in the
// hardware's 16-lane groups) and the staging writes of the 4 channel groups of one position land in different bank quads.
// Zero padding = LDS slots that are never written (zero-filled once); a displaced row that lies outside the image is all
// padding, its wave skips the arithmetic and stores zeros.  58 KB of LDS and <= 128 registers: two workgroups per CU, the next
// chunk's global loads fly (in registers) under the multiply-adds.  Offsets are arithmetic in (thread, item): no private arrays.
template <int W_>
__global__ void __launch_bounds__(VV_WG, 2)
correlation_nhwc_kernel(const float* __restrict__ f1, const float* __restrict__ f2, const int cs, const int C,
                        const int H, float* __restrict__ out, const int ocs, const int ocoff, const float slope,
                        const int NG) {
  constexpr int D = 21, MD = 20, CK = 16, NC4 = CK / 4;
  constexpr int DPW = 128 / W_;                 // row displacements per wave
  constexpr int DYB = 4 * DPW;                  // ... per workgroup
  constexpr int LPD = W_ / 2;                   // lanes per row displacement
  constexpr int NJ = W_ / 4;
  constexpr int QS = ((W_ + 2 * MD + 3) / 4 + 15) / 16 * 16;      // slots per pos-mod-4 plane of f2 (multiple of 16)
  constexpr int Q1 = (NJ + 15) / 16 * 16;
  constexpr int P2 = 4 * QS + 1, P1 = 4 * Q1 + 1;                 // float4 per channel-group plane (+1: bank shift)
  constexpr int F1SZ = NC4 * P1, F2SZ = NC4 * P2;                 // float4 per tile
  constexpr int PER = W_ * NC4;                                   // float4 items per staged row and chunk
  constexpr int NIT = (1 + DYB) * PER / VV_WG;
  static_assert(((1 + DYB) * PER) % VV_WG == 0 && (PER % VV_WG == 0 || VV_WG % PER == 0), "item split");
  extern __shared__ float4 cl[];
  const float4* F1 = cl;
  const float4* F2 = cl + F1SZ;

  const int y = blockIdx.x / NG, grp = blockIdx.x % NG, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane % LPD, j = li % NJ, par = li / NJ;
  const int slot = wave * DPW + lane / LPD;
  const int ti = grp * DYB + slot;              // row displacement index of this thread
  const float* p1 = f1 + (int64_t)b * H * W_ * cs;
  const float* p2 = f2 + (int64_t)b * H * W_ * cs;

  for (int e = tid; e < F1SZ + DYB * F2SZ; e += VV_WG) cl[e] = make_float4(0.f, 0.f, 0.f, 0.f);

  // staging item k of this thread: it = tid + k * 256 -> staged row (k * 256) / PER (0 = f1, 1.. = displaced f2 row), element
  // rem = tid + (k * 256) % PER of that row: channel group rem % NC4 of column rem / NC4
  float4 r[NIT];
  auto rowvalid = [&](const int row) -> bool {          // uniform over the workgroup
    if (row == 0) return true;
    const int t2 = grp * DYB + row - 1;
    const int yy = y + 2 * (t2 - MD / 2);
    return t2 < D && (unsigned)yy < (unsigned)H;
  };
  // branch-free staging: a displaced row outside the image is loaded from a clamped (valid) address and zeroed by a select when
  // it is committed -- per-row branches made the compiler re-roll the rows into a loop that indexes r[] dynamically (scratch)
  auto issue = [&](const int c0) {
    vv_static_for<0, NIT>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int row = (k * VV_WG) / PER;
      const int rem = tid + (k * VV_WG) % PER;
      const int c4 = rem % NC4, x = rem / NC4;
      int yy = row == 0 ? y : y + 2 * (grp * DYB + row - 1 - MD / 2);
      yy = min(max(yy, 0), H - 1);
      const float* src = (row == 0 ? p1 : p2) + ((int64_t)yy * W_ + x) * cs + c4 * 4 + c0;
      r[k] = *reinterpret_cast<const float4*>(src);
    });
  };
  auto commit = [&]() {
    vv_static_for<0, NIT>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int row = (k * VV_WG) / PER;
      const int rem = tid + (k * VV_WG) % PER;
      const int c4 = rem % NC4, x = rem / NC4;
      const int pos = x + MD;
      const int off = row == 0 ? c4 * P1 + (x & 3) * Q1 + (x >> 2)
                               : F1SZ + (row - 1) * F2SZ + c4 * P2 + (pos & 3) * QS + (pos >> 2);
      const bool ok = rowvalid(row);
      float4 v = r[k];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      cl[off] = v;
    });
  };

  v2f aa[D], ab[D];            // .x: channels 0,1 mod 4 pairs ... the two lanes of a packed accumulator are added at the end
#pragma unroll
  for (int k = 0; k < D; ++k) { aa[k] = (v2f){0.f, 0.f}; ab[k] = (v2f){0.f, 0.f}; }
  const float4* q1 = F1 + par * Q1 + j;
  const float4* q2 = F2 + slot * F2SZ + par * QS + j;
  // does this WAVE have anything but padding to multiply?  (one or two row displacements per wave)
  bool work = false;
#pragma unroll
  for (int d = 0; d < DPW; ++d) work = work || rowvalid(1 + wave * DPW + d);

  issue(0);
  for (int c0 = 0; c0 < C; c0 += CK) {
    __syncthreads();                       // previous chunk fully consumed (first pass: zero fill done)
    commit();
    __syncthreads();
    if (c0 + CK < C) issue(c0 + CK);
    if (work) {
      // per channel group: 6 window blocks (4, 4, 4, 4, 4, 2 positions) in a software pipeline -- the reads of block g+1 are
      // issued before the multiply-adds of block g, and a scheduling fence per block keeps the compiler from hoisting the whole
      // window (88 registers): two blocks (32 registers) are live at any time
      constexpr int NB = (D + 1 + 3) / 4;
#pragma unroll 1
      for (int c4 = 0; c4 < NC4; ++c4) {
        const float4* w2 = q2 + c4 * P2;
        const float4 va = q1[c4 * P1], vb = q1[c4 * P1 + 2 * Q1];      // columns x = 4j+par and x+2
        const v2f valo = {va.x, va.y}, vahi = {va.z, va.w}, vblo = {vb.x, vb.y}, vbhi = {vb.z, vb.w};
        float4 win[2][4];
        auto load_block = [&](auto gc) {
          constexpr int gi = decltype(gc)::value;
          vv_static_for<0, 4>([&](auto uc) {
            constexpr int m = gi * 4 + decltype(uc)::value;
            if constexpr (m <= D) win[gi & 1][m & 3] = w2[2 * (m & 1) * QS + (m >> 1)];      // f2 position x - 20 + 2m
          });
        };
        load_block(std::integral_constant<int, 0>{});
        vv_static_for<0, NB>([&](auto gc) {
          constexpr int gi = decltype(gc)::value;
          if constexpr (gi + 1 < NB) load_block(std::integral_constant<int, gi + 1>{});
          vv_static_for<0, 4>([&](auto uc) {
            constexpr int m = gi * 4 + decltype(uc)::value;
            if constexpr (m <= D) {
              // packed multiply-adds (v_pk_fma_f32): the kernel is bound by VALU ISSUE (one instruction per 4 cycles and wave,
              // one or two waves per SIMD), a packed instruction retires two of the four channels per issue slot
              const float4 v = win[gi & 1][m & 3];
              const v2f vlo = {v.x, v.y}, vhi = {v.z, v.w};
              if constexpr (m < D) aa[m] = __builtin_elementwise_fma(vahi, vhi, __builtin_elementwise_fma(valo, vlo, aa[m]));
              if constexpr (m > 0) ab[m - 1] = __builtin_elementwise_fma(vbhi, vhi, __builtin_elementwise_fma(vblo, vlo, ab[m - 1]));
            }
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    }
  }
  if (ti < D) {
    const float scale = 1.f / (float)C;
    const int x = 4 * j + par;
    float* o = out + ((int64_t)(b * H + y) * W_ + x) * ocs + ocoff + ti * D;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      float v = (aa[k].x + aa[k].y) * scale;
      o[k] = v > 0.f ? v : v * slope;
      v = (ab[k].x + ab[k].y) * scale;
      o[2 * ocs + k] = v > 0.f ? v : v * slope;
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

// Correlation with kernel_size > 1 (correlation_cuda_kernel.cu:34-106 in full generality: the k x k patch sum around both
// positions).  FlowNet2 never uses it (FlowNetC.py:24-30 has kernel_size 1), so this is the plain form: one thread per output
// element, NCHW, zero padding by predicate; consecutive threads = consecutive x.
__global__ void __launch_bounds__(VV_WG)
correlation_generic_kernel(const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out, const int B,
                           const int C, const int H, const int W, const int oC, const int oH, const int oW, const int pad,
                           const int ksz, const int md, const int s1, const int s2) {
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  const int64_t n = (int64_t)B * oC * oH * oW;
  if (e >= n) return;
  const int ox = (int)(e % oW);
  const int oy = (int)((e / oW) % oH);
  const int tc = (int)((e / ((int64_t)oW * oH)) % oC);
  const int b = (int)(e / ((int64_t)oW * oH * oC));
  const int kr = (ksz - 1) / 2, dr = md / s2, D = 2 * dr + 1;
  const int tj = tc / D - dr, ti = tc % D - dr;
  const int y1 = oy * s1 + md + kr - pad, x1 = ox * s1 + md + kr - pad;      // un-padded coordinates of the patch centres
  const int y2 = y1 + tj * s2, x2 = x1 + ti * s2;
  const int64_t HW = (int64_t)H * W;
  float acc = 0.f;
  for (int j = -kr; j <= kr; ++j)
    for (int i = -kr; i <= kr; ++i) {
      const int ya = y1 + j, xa = x1 + i, yb = y2 + j, xb = x2 + i;
      if ((unsigned)ya >= (unsigned)H || (unsigned)xa >= (unsigned)W || (unsigned)yb >= (unsigned)H || (unsigned)xb >= (unsigned)W)
        continue;                                                             // one factor is zero padding
      const float* pa = in1 + (int64_t)b * C * HW + (int64_t)ya * W + xa;
      const float* pb = in2 + (int64_t)b * C * HW + (int64_t)yb * W + xb;
      for (int c = 0; c < C; ++c) acc = fmaf(pa[c * HW], pb[c * HW], acc);
    }
  out[e] = acc / (float)(ksz * ksz * C);
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
  if (corr_type_multiply != 1 || kernel_size < 1 || kernel_size % 2 == 0) return VV_ERR_UNSUPPORTED;   // the reference's kernel multiplies
  int oC, oH, oW;
  int rc = vv_correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oC, &oH, &oW);
  if (rc) return rc;
  if (kernel_size > 1) {            // general patch correlation: the plain kernel (FlowNet2 itself only uses kernel_size 1)
    const int64_t n = (int64_t)B * oC * oH * oW;
    VV_LAUNCH(correlation_generic_kernel, dim3((unsigned)((n + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, (hipStream_t)stream, in1, in2, out,
              B, C, H, W, oC, oH, oW, pad_size, kernel_size, max_displacement, stride1, stride2);
    VV_CHECK_LAUNCH();
    return VV_OK;
  }
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

template <int W_>
static int launch_corr_nhwc(const float* f1, const float* f2, int cs, int B, int C, int H, float* out, int ocs, int ocoff,
                            float slope, hipStream_t st) {
  constexpr int DYB = 4 * (128 / W_), NC4 = 4;          // 16-channel chunks: the kernel's CK / 4
  constexpr int QS = ((W_ + 40 + 3) / 4 + 15) / 16 * 16, Q1 = (W_ / 4 + 15) / 16 * 16;
  constexpr size_t bytes = (size_t)(NC4 * (4 * Q1 + 1) + DYB * NC4 * (4 * QS + 1)) * 16;
  {   // idempotent and cheap: set on every call rather than remembering it in a static
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(correlation_nhwc_kernel<W_>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return VV_HIP_STATUS(e);
  }
  const int NG = (21 + DYB - 1) / DYB;
  VV_LAUNCH(correlation_nhwc_kernel<W_>, dim3(H * NG, B), dim3(VV_WG), bytes, st, f1, f2, cs, C, H, out, ocs, ocoff,
            slope, NG);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_correlation_nhwc(const float* f1, const float* f2, int32_t cstride, int32_t B, int32_t C, int32_t H,
                                   int32_t W, float* out, int32_t out_cstride, int32_t out_coff, float slope,
                                   vv_stream stream) {
  if (!f1 || !f2 || !out || B <= 0 || H <= 0) return VV_ERR_BAD_ARG;
  if (C % 16 || cstride % 4 || cstride < C) return VV_ERR_UNSUPPORTED;
  if (((uintptr_t)f1 | (uintptr_t)f2) & 15) return VV_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (W == 128) return launch_corr_nhwc<128>(f1, f2, cstride, B, C, H, out, out_cstride, out_coff, slope, st);
  if (W == 64) return launch_corr_nhwc<64>(f1, f2, cstride, B, C, H, out, out_cstride, out_coff, slope, st);
  return VV_ERR_UNSUPPORTED;   /* other widths: vv_correlation_fwd (generic, NCHW) */
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
