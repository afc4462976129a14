// Winograd F(2x2, 3x3) form of FlowNet2's large stride-1 3x3 convolutions (misc.py:8-28 conv(k=3, s=1) + LeakyReLU(0.1);
// FlowNetSD.py:9-103 conv1_1 / conv2_1 / conv3_1 ..., FlowNetFusion.py:9-64 conv1_1, FlowNetC.py conv3_1 ...): the same
// arithmetic as wino_conv_kernel (vv_wino.hip) -- Y = A^T [(G g G^T) . (B^T d B)] A in fp32 on v_mfma_f32_32x32x2_f32, 16 instead
// of 36 matrix-core K steps per 2x2 outputs -- for images of any size H x W with H even and W % 32 == 0 (every pyramid level of
// a 64-aligned frame from H/2 down to H/32), plain NHWC input with any channel count (K zero-padded to 8 in the transformed
// panel; the activation buffers are ceil4(C) wide and finite, vec_vad_amd/flownet2.py::_Buf), bias + LeakyReLU epilogue, output
// into a channel slice of the consumer's concat buffer.  Round 3 ran these layers on conv2d_mfma_kernel at 95 - 105 TFLOP/s
// (0.6 - 0.67 of the fp32 MFMA peak, direct form): they fill the chip, so 2.25x fewer MFMAs is time, not idle CUs.
//
// One workgroup = 4 waves = 32 tiles = 4 x 32 output pixels (2 tile rows x 16 tile columns) x 32 output channels; wave = xi, its
// four GEMMs (nu) in 64 accumulator registers; the transformed filter panel [xi*4+nu][Kp/8][2][N][4] (vv_pack_wino) goes L2 ->
// registers one 8-channel chunk ahead, the 6 x 34 halo tile global -> registers -> LDS one chunk ahead, odd / even columns in
// separate planes (vv_wino.hip has the derivation of every piece; this file is its run-time-geometry sibling without the
// UNet-specific parts: groups, BatchNorm on load / partial sums, concat sources).
#include <type_traits>
#include "vv_common.h"

namespace {

constexpr int WN = 256;
constexpr int SB_MASK = 0x386;         // may cross a scheduling barrier: VALU, SALU, LDS -- not MFMA, not VMEM
constexpr int TPI = 16, TROWS = 2;     // tile columns / rows of a workgroup
constexpr int HH = 2 * TROWS + 2, HW = 2 * TPI + 2, HWH = HW / 2;

__global__ void __launch_bounds__(WN, 3)
conv2d_wino_kernel(const float* __restrict__ src, const int src_cs, const int src_co, const int64_t src_elems,
                   const float* __restrict__ wpanel, const float* __restrict__ bias, const float slope,
                   float* __restrict__ out, const int out_cs, const int out_co,
                   const int B, const int H, const int W, const int CinP, const int Cout, const int NN, const int total, const int nper) {
  constexpr int CK = 8, S = CK + 4, S4 = S / 4, Q = CK / 4;
  constexpr int A4 = HH * HW * S4;
  constexpr int NITEMS = HH * HW * Q;
  constexpr int NIT = (NITEMS + WN - 1) / WN;
  constexpr int EX4 = 4 * 16 * 64 / 2;                     // epilogue exchange: [wave][16 regs][64 lanes] float2
  constexpr int L4 = A4 > EX4 ? A4 : EX4;
  __shared__ float4 lds4[L4];
  float* lds = reinterpret_cast<float*>(lds4);
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int nn = w % NN; w /= NN;      // the N tiles of one pixel block are neighbours in launch order: they share the halo in L2
  const int bx = W / (2 * TPI), by = (H + 2 * TROWS - 1) / (2 * TROWS);      // H % 4 == 2: the last block's second tile row is masked
  const int xb = w % bx; w /= bx;
  const int yb = w % by;
  const int b = w / by;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int y0 = yb * (2 * TROWS) - 1, x0 = xb * (2 * TPI) - 1;      // conv-input coordinates of halo (0, 0)
  const int KQ = CinP >> 3;
  const int co0 = nn * 32;

  // ---- halo staging
  float4 r[NIT];
  int slot[NIT];
  unsigned voff[NIT];
  const int q4 = (tid % Q) * 4;
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = tid + k * WN;
    const int q = it % Q, hp = it / Q;
    const int hx = hp % HW, hy = hp / HW;
    const bool inr = NITEMS % WN == 0 || it < NITEMS;
    const int y = y0 + hy, x = x0 + hx;
    const bool ok = inr && (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
    voff[k] = ok ? (unsigned)((((b * H + y) * W + x) * src_cs + src_co + q4) * 4) : 0x80000000u;
    slot[k] = inr ? (hy * HW + (hx & 1) * HWH + (hx >> 1)) * S4 + q : -1;
  }
  // num_records = the buffer's size: the K padding of the last pixel reads past the tensor -> zeros instead of a fault
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(src_elems * 4 < 0x7FFFFFFF ? src_elems * 4 : 0x7FFFFFFF), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpanel), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned bvo = (unsigned)(half * Cout + co0 + l31) * 16u;                // this lane inside a [2][Cout] float4 slab
  const int bnu = KQ * 2 * Cout * 16;                                            // bytes between nu slabs
  const int bxi = xi * 4 * bnu;
  auto issue = [&](const int c0) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[k], c0 * 4, 0);
      r[k] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (NITEMS % WN == 0 || k < NIT - 1 || slot[k] >= 0) lds4[slot[k]] = r[k];
  };
  auto load_u = [&](const int c0, const int n) -> v4f {
    return __builtin_amdgcn_raw_buffer_load_b128(rsW, bvo, bxi + n * bnu + (c0 >> 3) * 2 * Cout * 16, 0);
  };

  // ---- this lane's tile and its patch origin in LDS
  const int tyl = l31 / TPI, tx = l31 % TPI;
  const int pbase = ((2 * tyl) * HW + tx) * S4 + half;
  auto patch = [&](const int a, const int bcol) -> v4f {
    return ldsA[pbase + (a * HW + (bcol & 1) * HWH + (bcol >> 1)) * S4];
  };
  // B^T rows: xi 0: d0-d2   1: d1+d2   2: d2-d1   3: d1-d3   ->   R = d[a1] + sg * d[a2]
  const int a1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), a2 = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
  const float sg = xi == 1 ? 1.f : -1.f;

  v16f acc[4];
  v4f u[4];
  const int klast = CinP - CK;
  auto compute = [&](const int c0, const auto first, const auto more) {
    const int kn = c0 + CK;
    const int knext = kn < klast ? kn : klast;             // the last chunk re-reads its own taps (unused)
    if constexpr (decltype(more)::value) issue(kn);
    v4f R[4];
#pragma unroll
    for (int bc = 0; bc < 4; ++bc) {
      const v4f d1 = patch(a1, bc), d2 = patch(a2, bc);
      R[bc] = d1 + sg * d2;
    }
    v4f V[4];
    V[0] = R[0] - R[2];
    V[1] = R[1] + R[2];
    V[2] = R[2] - R[1];
    V[3] = R[1] - R[3];
    __builtin_amdgcn_sched_barrier(SB_MASK);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if constexpr (decltype(first)::value) {   // the accumulators start from the instruction's inline-constant 0
        const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].x, u[n].x, z, 0, 0, 0);
      } else {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].x, u[n].x, acc[n], 0, 0, 0);
      }
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].y, u[n].y, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].z, u[n].z, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].w, u[n].w, acc[n], 0, 0, 0);
      u[n] = load_u(knext, n);
      __builtin_amdgcn_sched_barrier(SB_MASK);
    }
  };

  issue(0);
#pragma unroll
  for (int n = 0; n < 4; ++n) u[n] = load_u(0, n);
  commit();
  __syncthreads();
  const std::true_type yes{};
  const std::false_type no{};
  if (CK >= CinP) {
    compute(0, yes, no);
  } else {
    compute(0, yes, yes);
    int c0 = CK;
    for (; c0 + CK < CinP; c0 += CK) {
      __syncthreads();                  // every wave finished reading the previous chunk
      commit();
      __syncthreads();
      compute(c0, no, yes);
    }
    __syncthreads();
    commit();
    __syncthreads();
    compute(c0, no, no);
  }

  // ---- epilogue.  Columns in registers:  T[0] = M0 + M1 + M2,  T[1] = M1 - M2 - M3  (M = this wave's xi, indexed by nu);
  //      rows across the four waves through LDS:  Y[0] = T(xi0) + T(xi1) + T(xi2),  Y[1] = T(xi1) - T(xi2) - T(xi3).
  //      Wave w finishes accumulator registers 4w .. 4w+3 = tiles 8w .. 8w+7 (both rows): 16 buffer stores per wave.
  __syncthreads();                      // all MFMA-phase LDS reads done: LDS becomes the exchange buffer
  v2f* ex2 = reinterpret_cast<v2f*>(lds) + lane;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const v2f t = {acc[0][i] + acc[1][i] + acc[2][i], acc[1][i] - acc[2][i] - acc[3][i]};
    ex2[(xi * 16 + i) * 64] = t;
  }
  __syncthreads();
  // accumulator register i of the MFMA result = tile row (i & 3) + 8 (i >> 2) + 4 half of the 32-tile block: wave xi takes registers
  // 4 xi .. 4 xi + 3, i.e. tiles 8 xi + j (half 0) and 8 xi + j + 4 (half 1), j = 0..3
  const float bv = bias ? bias[co0 + l31] : 0.f;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00020000);
  constexpr int LP = 8;                                          // pixels between the two lane halves (4 tiles to the right)
  const int vo = (half * LP * out_cs + out_co + co0 + l31) * 4;
  const v2f* exw = ex2 + xi * 4 * 64;
  const bool nok = co0 + l31 < Cout;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t2 = xi * 8 + j;                                   // wave-uniform part of the tile index
    const int oy = yb * (2 * TROWS) + 2 * (t2 / TPI), ox = xb * (2 * TPI) + 2 * (t2 % TPI);
    const v2f t0 = exw[(0 * 16 + j) * 64], t1 = exw[(1 * 16 + j) * 64], t2v = exw[(2 * 16 + j) * 64], t3 = exw[(3 * 16 + j) * 64];
    v2f ya = t0 + t1 + t2v + bv, yv = t1 - t2v - t3 + bv;
    // LeakyReLU (slope 1 = none): compare + select keeps a NaN
    ya[0] = ya[0] < 0.f ? ya[0] * slope : ya[0]; ya[1] = ya[1] < 0.f ? ya[1] * slope : ya[1];
    yv[0] = yv[0] < 0.f ? yv[0] * slope : yv[0]; yv[1] = yv[1] < 0.f ? yv[1] * slope : yv[1];
    const int so = (((b * H + oy) * W + ox) * out_cs) * 4;
    if (nok && oy < H) {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ya[0]), rsO, vo, so, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ya[1]), rsO, vo, so + out_cs * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[0]), rsO, vo, so + W * out_cs * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv[1]), rsO, vo, so + (W + 1) * out_cs * 4, 0);
    }
  }
}

}  // namespace

// FlowNet2 conv(k = 3, stride 1, pad 1) [+ LeakyReLU(slope)] in Winograd F(2x2, 3x3) form.  `panel` = vv_pack_wino(mode 0) of the
// nn.Conv2d weight [Cout][Cin][3][3] with K padded to CinP (multiple of 8), N = Cout (multiple of 32).  H % 2 == 0, W % 32 == 0.
// src_elems: number of floats in the source buffer (loads past it return zeros: the K padding of the last pixel).
extern "C" int vv_conv2d_wino(const float* src, int32_t src_cstride, int32_t src_coff, int64_t src_elems, const float* panel,
                              const float* bias, float slope, float* out, int32_t out_cstride, int32_t out_coff, int32_t B, int32_t H,
                              int32_t W, int32_t CinP, int32_t Cout, vv_stream stream) {
  if (!src || !panel || !out || B <= 0) return VV_ERR_BAD_ARG;
  if (H % 2 || W % 32 || CinP % 8 || Cout % 32 || src_cstride % 4 || src_coff % 4) return VV_ERR_UNSUPPORTED;
  // 32-bit byte offsets into the source / output buffers
  if ((int64_t)B * H * W * src_cstride * 4 >= (1ll << 31) || (int64_t)B * H * W * out_cstride * 4 >= (1ll << 31)) return VV_ERR_UNSUPPORTED;
  const int NN = Cout / 32;
  const int total = B * ((H + 3) / 4) * (W / 32) * NN;
  const int nper = (total + 7) / 8;
  VV_LAUNCH(conv2d_wino_kernel, dim3(nper * 8), dim3(WN), 0, (hipStream_t)stream, src, src_cstride, src_coff, src_elems, panel, bias,
            slope, out, out_cstride, out_coff, B, H, W, CinP, Cout, NN, total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
