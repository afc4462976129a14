// Winograd F(2x2, 3x3) convolution on the matrix cores (gfx950, v_mfma_f32_32x32x2_f32) for the UNet bank's 3x3 / stride 1 /
// pad 1 layers (model/unet.py:10,13) -- forward and data-gradient (the latter = the same convolution with the flipped,
// transposed filter, transformed at pack time).
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A     per 4x4 input patch d -> 2x2 outputs, summed over input channels
//
// The 16 element-wise products of a patch become 16 independent GEMMs  M[xi,nu] = V[xi,nu] (tiles x Cin) * U[xi,nu] (Cin x
// Cout): 16 MFMA-K steps per 4 output pixels instead of 36 for the direct form (2.25x fewer matrix-core cycles).  All in fp32;
// the transforms only add / subtract and halve, so the result differs from the direct convolution by a few ulp.
//
// One workgroup = 256 threads = 4 waves = 64 tiles (256 output pixels) x 32 output channels of one UNet, two workgroups
// per CU (<= 256 registers per lane) so that one's staging / epilogue phases run under the other's MFMAs.
//   wave = (tile group tg = 0..1 of 32 tiles) x (xi half xh): its 8 GEMMs (xi in {2xh, 2xh+1}, nu = 0..3) live in 128
//   accumulator registers; lane l owns tile l&31, channel half l>>5 (A operand) / output channel l&31 (B operand, result).
//   Input halo tile [NI][HH][HW][8+4] and the chunk's transformed filter panel [16][2][32] float4 go
//   global -> registers -> LDS one 8-channel chunk ahead (the producer's BatchNorm+ReLU is applied on the way in, like in
//   vv_conv.hip); odd and even columns are stored in separate planes so that lanes walking tile columns read consecutive
//   slots.  V is built in registers from 8 patch reads per xi (row transform, then column transform) under the MFMAs.
//   Epilogue: each wave applies A^T . A to its xi half, the halves meet in LDS, then bias, NHWC store and the BatchNorm
//   sum / sum-of-squares partials exactly like the direct kernel.
#include "vv_common.h"
// VV_EXP (compile-time, default 0): elimination switches used to find where the time goes (profiles/README.md, round 2) -- every
// value other than 0 computes WRONG results: 1 = no MFMAs, 2 = no input transform, 3 = LDS commit of the first chunk only,
// 4 = global loads of the first chunk only, 5 = no barriers.
#ifndef VV_EXP
#define VV_EXP 0
#endif

namespace {

constexpr int NTG = 2;                 // tile groups (of 32 tiles) per workgroup
constexpr int WN = NTG * 2 * 64;       // threads per workgroup: (tile group) x (xi half) waves
constexpr int WT = NTG * 32;           // tiles per workgroup

template <int H_>
struct WGeo {
  static constexpr int TPI = H_ / 2;                       // tiles per image side
  static constexpr int TP = TPI * TPI;                     // tiles per image
  static constexpr int TPW = TP < WT ? TP : WT;          // tiles of one image handled by one workgroup
  static constexpr int NI = WT / TPW;                     // images per workgroup
  static constexpr int PARTS = TP / TPW;                   // workgroups per image
  static constexpr int TROWS = TPW / TPI;                  // tile rows per part
  static constexpr int HH = 2 * TROWS + 2, HW = H_ + 2;
};

template <int H_>
__global__ void __launch_bounds__(WN, 2)
wino_conv_kernel(const vv_conv_params p, const int NT, const int NN, const int total, const int nper) {
  using G_ = WGeo<H_>;
  constexpr int TPI = G_::TPI, TPW = G_::TPW, NI = G_::NI, PARTS = G_::PARTS, HH = G_::HH, HW = G_::HW, HWH = HW / 2;
  constexpr int CK = 8, S = CK + 4, S4 = S / 4, Q = CK / 4;
  constexpr int A4 = NI * HH * HW * S4;
  constexpr int B4 = 16 * 2 * 32;                          // [xi*4+nu][half][co] float4
  constexpr int NITEMS = NI * HH * HW * Q;
  constexpr int NIT = (NITEMS + WN - 1) / WN;
  constexpr int NBT = B4 / WN;
  constexpr int EX4 = NTG * 4 * 16 * 64 / 4;               // epilogue exchange: [tg][2x2][16 regs][64 lanes] floats
  constexpr int LDS4 = (A4 + B4) > EX4 ? (A4 + B4) : EX4;
  static_assert(B4 % WN == 0 && WN % Q == 0, "staging geometry");
  __shared__ float4 lds4[LDS4];
  float* lds = reinterpret_cast<float*>(lds4);
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);
  const v4f* ldsB = reinterpret_cast<const v4f*>(lds4) + A4;

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int pt = w % NT; w /= NT;
  const int nn = w % NN;
  const int g = w / NN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int tg = wave >> 1, xh = wave & 1;
  const int img0 = (pt / PARTS) * NI, part = pt % PARTS;
  const int y0 = part * (2 * G_::TROWS) - 1;               // conv-input row of halo row 0 (column origin is -1)
  const VVSrc s = vv_make_src(p, g, H_, H_);
  const int Cout = p.Cout, CinP = p.CinP, KQ = CinP >> 3;
  const int co0 = nn * 32;
  const float* __restrict__ wg = p.w + (int64_t)g * p.w_gstride;

  // ---- staging set-up (once per workgroup)
  float4 r[NIT];
  int pix[NIT];          // source pixel offset relative to the tile origin, < 0: never valid
  short hyv[NIT], imv[NIT];
  int slot[NIT];         // destination float4 index in LDS
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = tid + k * WN;
    const int q = it % Q, hp = it / Q;
    const int hx = hp % HW, t = hp / HW;
    hyv[k] = (short)(t % HH);
    imv[k] = (short)(t / HH);
    const int x = hx - 1;
    const bool ok = (NITEMS % WN == 0 || it < NITEMS) && (unsigned)x < (unsigned)H_;
    pix[k] = ok ? (imv[k] * H_ + hyv[k]) * H_ + hx : -(1 << 30);
    slot[k] = (NITEMS % WN == 0 || it < NITEMS) ? ((imv[k] * HH + hyv[k]) * HW + (hx & 1) * HWH + (hx >> 1)) * S4 + q : -1;
  }
  unsigned boff[NBT];
  float4 rb[NBT];
#pragma unroll
  for (int k = 0; k < NBT; ++k) {
    const int it = tid + k * WN;
    const int col = it & 31, row = it >> 5;                // row = xinu*2 + half
    boff[k] = (unsigned)((((row >> 1) * KQ) * 2 + (row & 1)) * Cout + co0 + col) * 16u;
  }
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, 0x7FFFFFFF, 0x00020000);
  float4 sa, sb;
  bool act = false;
  // Everything about an item that does not depend on the chunk is computed ONCE: its validity (image row / image / column inside
  // the tensor) and its byte offset inside the source tensor; a chunk only moves the scalar offset of the buffer load.  A
  // concat input (VV_IN_CAT) switches to its second tensor at csplit: the offsets are rebuilt there (once per workgroup).
  unsigned valid = 0;
  unsigned voff[NIT];
  const int tile = (img0 * H_ + y0) * H_ - 1;
  const int q4 = (tid % Q) * 4;
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int y = y0 + hyv[k];
    const bool ok = (unsigned)y < (unsigned)H_ && (img0 + imv[k]) < s.B && pix[k] >= 0;
    valid |= ok ? (1u << k) : 0u;
  }
  __amdgpu_buffer_rsrc_t rs;
  int cur_second = -1, soff0 = 0;
  auto set_source = [&](const bool second) {
    const float* base = second ? s.p1 + s.co1 : s.p0 + s.co0;
    const int cs = second ? s.cs1 : s.cs0;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
    soff0 = second ? -s.csplit * 4 : 0;
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      voff[k] = ((valid >> k) & 1u) ? (unsigned)((tile + pix[k]) * cs + q4) * 4u : 0x80000000u;
    cur_second = second ? 1 : 0;
  };
  auto issue = [&](const int c0) {
    const int c = c0 + q4;
    act = (s.mode == VV_IN_ACT) || (s.mode == VV_IN_CAT && c < s.csplit);
    if (act) {
      sa = *reinterpret_cast<const float4*>(s.a + c);
      sb = *reinterpret_cast<const float4*>(s.b + c);
    }
    const int second = __builtin_amdgcn_readfirstlane((int)((s.mode == VV_IN_CAT) && c0 >= s.csplit));
    if (second != cur_second) set_source(second != 0);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[k], c0 * 4 + soff0, 0);
      r[k] = make_float4(v.x, v.y, v.z, v.w);
    }
    const int so = (c0 >> 3) * 2 * Cout * 16;
#pragma unroll
    for (int k = 0; k < NBT; ++k) {
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rsW, boff[k], so, 0);
      rb[k] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  auto commit = [&](const int bo) {                        // bo: float4 offset of the destination staging buffer
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (NITEMS % WN == 0 || k < NIT - 1 || slot[k] >= 0) {      // only the last item of a ragged item count can be void
        float4 v = r[k];
        if (act && ((valid >> k) & 1u)) v = vv_act4(v, sa, sb);
        lds4[bo + slot[k]] = v;
      }
#pragma unroll
    for (int k = 0; k < NBT; ++k) lds4[bo + A4 + tid + k * WN] = rb[k];
  };

  // ---- this lane's tile and its patch origin in LDS
  const int tt = tg * 32 + l31;
  const int tim = tt / TPW, trem = tt % TPW;
  const int tyl = trem / TPI, tx = trem % TPI;
  // patch pixel (a, b): halo row 2*tyl + a, halo column 2*tx + b -> plane (b & 1), slot tx + (b >> 1)
  const int pbase = ((tim * HH + 2 * tyl) * HW + tx) * S4 + half;
  auto patch = [&](const int bo, const int a, const int b) -> v4f {
    return ldsA[bo + pbase + (a * HW + (b & 1) * HWH + (b >> 1)) * S4];
  };

  v16f acc[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[x][n][i] = 0.f;

  auto compute = [&](const int bo) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      // B^T rows: xi 0: d0-d2   1: d1+d2   2: d2-d1   3: d1-d3   ->   R = d[a1] + sg * d[a2]   (wave-uniform a1, a2, sg)
      const int xi = 2 * xh + x;
      const int a1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), a2 = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
      const float sg = xi == 1 ? 1.f : -1.f;
      v4f R[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#if VV_EXP == 2
        R[b] = patch(bo, a1, b);
#else
        const v4f d1 = patch(bo, a1, b), d2 = patch(bo, a2, b);
        R[b] = d1 + sg * d2;
#endif
      }
      v4f V[4];
#if VV_EXP == 2
      V[0] = R[0]; V[1] = R[1]; V[2] = R[2]; V[3] = R[3];
#else
      V[0] = R[0] - R[2];
      V[1] = R[1] + R[2];
      V[2] = R[2] - R[1];
      V[3] = R[1] - R[3];
#endif
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const v4f u = ldsB[bo + ((xi * 4 + n) * 2 + half) * 32 + l31];
#if VV_EXP == 1
        acc[x][n][0] += V[n].x * u.x + V[n].y * u.y + V[n].z * u.z + V[n].w * u.w;
#else
        acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].x, u.x, acc[x][n], 0, 0, 0);
        acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].y, u.y, acc[x][n], 0, 0, 0);
        acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].z, u.z, acc[x][n], 0, 0, 0);
        acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].w, u.w, acc[x][n], 0, 0, 0);
#endif
      }
    }
  };

  issue(0);
  for (int c0 = 0; c0 < CinP; c0 += CK) {
#if VV_EXP != 5
    if (c0) __syncthreads();            // every wave finished reading the previous chunk
#endif
#if VV_EXP == 3
    if (c0 == 0)
#endif
    commit(0);
#if VV_EXP != 5
    __syncthreads();
#endif
#if VV_EXP == 4
    if (false)
#endif
    if (c0 + CK < CinP) issue(c0 + CK);
    compute(0);
  }

  // ---- epilogue.  Output transform of this wave's xi half:  T[x][q] = sum_nu M[x][nu] A[nu][q],  A^T = [1 1 1 0; 0 1 -1 -1]
  //      xh = 0 (xi 0,1): Y[0][q] = T[0][q] + T[1][q],  Y[1][q] = T[1][q]
  //      xh = 1 (xi 2,3): Y[0][q] = T[0][q],            Y[1][q] = -T[0][q] - T[1][q]        (T indexed by local x)
  __syncthreads();                      // all MFMA-phase LDS reads done: LDS becomes the exchange buffer
  float* ex = lds + (tg * 4) * 16 * 64 + lane;
  float bias = 0.f, s1 = 0.f, s2 = 0.f;
  const bool relu = (p.pad0 & VV_CONV_RELU) != 0;
  if (xh == 0 && p.bias) bias = p.bias[(int64_t)g * p.bias_gstride + co0 + l31];
  float y4[16][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float t00 = acc[0][0][i] + acc[0][1][i] + acc[0][2][i], t01 = acc[0][1][i] - acc[0][2][i] - acc[0][3][i];
    const float t10 = acc[1][0][i] + acc[1][1][i] + acc[1][2][i], t11 = acc[1][1][i] - acc[1][2][i] - acc[1][3][i];
    if (xh == 0) {
      y4[i][0] = t00 + t10; y4[i][1] = t01 + t11; y4[i][2] = t10; y4[i][3] = t11;
    } else {
      y4[i][0] = t00; y4[i][1] = t01; y4[i][2] = -t00 - t10; y4[i][3] = -t01 - t11;
    }
  }
  if (xh == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int pq = 0; pq < 4; ++pq) ex[(pq * 16 + i) * 64] = y4[i][pq];
  }
  __syncthreads();
  float* __restrict__ outg = p.out.ptr + (int64_t)g * p.out.gstride + p.out.coff;
  const int ocs = p.out.cstride;
  if (xh == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * half;       // tile index inside the wave's group of 32
      const int t2 = tg * 32 + row;
      const int im = t2 / TPW, rem = t2 % TPW;
      const int oy = 2 * (part * G_::TROWS + rem / TPI), ox = 2 * (rem % TPI);
      const int img = img0 + im;
      if (img < p.B) {
        float* o = outg + ((int64_t)(img * H_ + oy) * H_ + ox) * ocs + co0 + l31;
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
          float v = y4[i][pq] + ex[(pq * 16 + i) * 64] + bias;
          if (relu) v = fmaxf(v, 0.f);          // VV_CONV_RELU: BatchNorm folded into the filter (eval mode), ReLU here
          o[((pq >> 1) * H_ + (pq & 1)) * ocs] = v;
          s1 += v; s2 = fmaf(v, v, s2);
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (xh == 0 && half == 0) {
      lds[tg * 32 + l31] = s1;
      lds[NTG * 32 + tg * 32 + l31] = s2;
    }
    __syncthreads();
    if (tid < 32) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < NTG; ++k) {
        t1 += lds[k * 32 + tid];
        t2 += lds[NTG * 32 + k * 32 + tid];
      }
      float* st = p.stats + ((int64_t)(g * NT + pt) * 2) * Cout + co0 + tid;
      st[0] = t1;
      st[Cout] = t2;
    }
  }
}

// U = G g G^T of every (ci, co) filter, in the B-operand panel layout [xi*4+nu][Kp/8][2][N][4].
//   G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
// mode 0: forward       g[a][b] = W[co = n][ci = k][a][b]
// mode 1: data gradient g[a][b] = W[co = k][ci = n][2-a][2-b]
__global__ void __launch_bounds__(VV_WG)
wino_pack_kernel(const vv_pack_entry* __restrict__ table, const float* __restrict__ params, const int64_t params_gstride,
                 float* __restrict__ packed, const int64_t packed_gstride) {
  // one workgroup = one 8-channel K group (kq) x 32 output channels: 256 (k, n) filters, one per thread.  Reads run along k
  // (8 filters = 288 contiguous bytes per n), the 16 transformed taps are exchanged through LDS so that every tap leaves as
  // two contiguous 512-byte runs [half][n][4].
  __shared__ float ex[16][256];
  const vv_pack_entry e = table[blockIdx.y];
  const int g = blockIdx.z;
  const float* src = params + (int64_t)g * params_gstride + e.src_off;
  float* dst = packed + (int64_t)g * packed_gstride + e.dst_off;
  const int KQ = e.KP >> 3, NB = e.N >> 5;
  const int t = threadIdx.x, kl = t & 7, nl = t >> 3;
  for (int blk = blockIdx.x; blk < KQ * NB; blk += gridDim.x) {
    const int kq = blk / NB, nb = blk % NB;
    const int k = kq * 8 + kl, n = nb * 32 + nl;
    float gk[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float v = 0.f;
        if (k < e.K) v = e.mode == 0 ? src[((int64_t)n * e.K + k) * 9 + a * 3 + b]
                                     : src[((int64_t)k * e.N + n) * 9 + (2 - a) * 3 + (2 - b)];
        gk[a][b] = v;
      }
    float r[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      r[0][b] = gk[0][b];
      r[1][b] = 0.5f * (gk[0][b] + gk[1][b] + gk[2][b]);
      r[2][b] = 0.5f * (gk[0][b] - gk[1][b] + gk[2][b]);
      r[3][b] = gk[2][b];
    }
    const int slot = (kl >> 2) * 128 + nl * 4 + (kl & 3);          // [half][n][j]
    __syncthreads();                                               // previous block's exchange fully read
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      ex[xi * 4 + 0][slot] = r[xi][0];
      ex[xi * 4 + 1][slot] = 0.5f * (r[xi][0] + r[xi][1] + r[xi][2]);
      ex[xi * 4 + 2][slot] = 0.5f * (r[xi][0] - r[xi][1] + r[xi][2]);
      ex[xi * 4 + 3][slot] = r[xi][2];
    }
    __syncthreads();
    const int hf = t >> 7, rem = t & 127;
#pragma unroll
    for (int xn = 0; xn < 16; ++xn)
      dst[((((int64_t)xn * KQ + kq) * 2 + hf) * e.N + nb * 32) * 4 + rem] = ex[xn][t];
  }
}

template <int H_>
int launch_wino(const vv_conv_params* p, hipStream_t st) {
  using G_ = WGeo<H_>;
  const int NT = ((p->B + G_::NI - 1) / G_::NI) * G_::PARTS;
  const int NN = p->Cout / 32;
  const int total = p->G * NN * NT;
  const int nper = (total + 7) / 8;
  VV_LAUNCH((wino_conv_kernel<H_>), dim3(nper * 8), dim3(WN), 0, st, *p, NT, NN, total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

}  // namespace

extern "C" int vv_wino_ntiles(int32_t B, int32_t H) {
  switch (H) {
    case 32: return ((B + WGeo<32>::NI - 1) / WGeo<32>::NI) * WGeo<32>::PARTS;
    case 16: return ((B + WGeo<16>::NI - 1) / WGeo<16>::NI) * WGeo<16>::PARTS;
    case 8: return ((B + WGeo<8>::NI - 1) / WGeo<8>::NI) * WGeo<8>::PARTS;
    case 4: return ((B + WGeo<4>::NI - 1) / WGeo<4>::NI) * WGeo<4>::PARTS;
  }
  return -1;
}

extern "C" int vv_conv_wino(const vv_conv_params* p, vv_stream stream) {
  if (!p || !p->src0.ptr || !p->w || !p->out.ptr) return VV_ERR_BAD_ARG;
  if (p->G <= 0 || p->B <= 0 || p->kind != VV_CONV3 || p->H != p->W) return VV_ERR_BAD_ARG;
  if (p->Cout % 32 || p->CinP % 8) return VV_ERR_UNSUPPORTED;
  if (p->in_mode == VV_IN_POOL || p->in_mode == VV_IN_CUBE)
    return VV_ERR_UNSUPPORTED;    // feed the materialised tensor (vv_pool_act / vv_cube_erase) as VV_IN_PLAIN
  hipStream_t st = (hipStream_t)stream;
  switch (p->H) {
    case 32: return launch_wino<32>(p, st);
    case 16: return launch_wino<16>(p, st);
    case 8: return launch_wino<8>(p, st);
    case 4: return launch_wino<4>(p, st);
  }
  return VV_ERR_UNSUPPORTED;
}

extern "C" int vv_pack_wino(const vv_pack_entry* table_dev, int32_t nentries, int32_t G, const float* params,
                            int64_t params_gstride, float* packed, int64_t packed_gstride, int32_t max_kn,
                            vv_stream stream) {
  if (!table_dev || !params || !packed || nentries <= 0) return VV_ERR_BAD_ARG;
  int bx = (max_kn + VV_WG - 1) / VV_WG;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  VV_LAUNCH(wino_pack_kernel, dim3(bx, nentries, G), dim3(VV_WG), 0, (hipStream_t)stream, table_dev, params, params_gstride,
            packed, packed_gstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
