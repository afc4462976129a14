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
// One workgroup = 256 threads = 4 waves = 32 tiles (128 output pixels) x 32 output channels of one UNet; wave = xi (0..3), its
// four GEMMs (nu = 0..3) live in 64 accumulator registers, so three workgroups fit a CU (<= 168 registers per lane) and a SIMD
// has a third wave to run while two wait.  Lane l owns tile l&31, channel half l>>5 (A operand) / output channel l&31 (B
// operand, result).
//   No two waves of a workgroup use the same filter taps, so the transformed filter never touches LDS: each lane loads its four
//   float4 per 8-channel chunk straight from the packed panel (L2-resident), one chunk ahead, into the registers the previous
//   chunk's MFMAs have just released.
//   The input halo tile [NI][HH][HW][8+4] goes global -> registers -> LDS one 8-channel chunk ahead (the producer's
//   BatchNorm+ReLU is applied on the way in, like in vv_conv.hip); odd and even columns are stored in separate planes so that
//   lanes walking tile columns read consecutive slots.  V is built in registers from 8 patch reads (row transform, then column
//   transform).
//   Epilogue: each wave applies the column half of A^T . A to its xi; the four waves meet in LDS and each finishes 8 tiles:
//   bias, NHWC buffer stores (wave-uniform offsets in scalar registers) and the BatchNorm sum / sum-of-squares partials like the
//   direct kernel.
// History (profiles/README.md, round 2): the first form held two xi per wave (128 accumulators, 64 tiles, filter panel staged
// through LDS, two workgroups per CU); this one is 11 % faster over the 27 launches of a Net4 step.
#include <cstdlib>
#include <type_traits>
#include "vv_common.h"
// VV_RING_D: prefetch distance of wino_ring_kernel.  (The elimination builds profiles/README.md quotes -- kernels with the MFMAs, the loads,
// the stores ... compiled out -- are not part of the shipped text: they are at git 536a9ae, switches VV_EXPM / VV_EXPR / VV_EXPC / VV_EXPG / VV_EXP4.)
#ifndef VV_RING_D
#define VV_RING_D 4
#endif
#ifndef VV_RING_MIN
#define VV_RING_MIN (4 * 512)      // work items (tiles x N tiles x UNets) from which the ring kernel takes a launch: runs of >= 4 per workgroup
#endif
namespace {

constexpr int WN = 256;                // threads per workgroup: one wave per xi
constexpr int WT = 32;                 // tiles per workgroup
constexpr int SB_MASK = 0x386;         // may cross a scheduling barrier: VALU, SALU, LDS -- not MFMA, not VMEM

// LDS hand-over between the waves of a workgroup WITHOUT the vector-memory drain a __syncthreads() can carry (wino_ring_kernel keeps
// LDS-DMAs in flight across it); the empty asm keeps the compiler from lifting later LDS reads above the barrier
__device__ __forceinline__ void vv_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int H_>
struct WGeo {
  static constexpr int TPI = H_ / 2;                       // tiles per image side
  static constexpr int TP = TPI * TPI;                     // tiles per image
  static constexpr int TPW = TP < WT ? TP : WT;            // tiles of one image handled by one workgroup
  static constexpr int NI = WT / TPW;                      // images per workgroup
  static constexpr int PARTS = TP / TPW;                   // workgroups per image
  static constexpr int TROWS = TPW / TPI;                  // tile rows per part
  static constexpr int HH = 2 * TROWS + 2, HW = H_ + 2;
};

// NB = N tiles (32 output channels each) per workgroup.  NB = 2 (round 5): a wave keeps eight GEMMs (128 accumulators, two workgroups per
// CU) and the input transform, the patch reads and the halo staging of a chunk serve 32 MFMAs instead of 16 -- an fp32 MFMA holds the
// SIMD's issue port, so a tile costs 64 cycles per MFMA plus ~5 per other instruction of every wave: fewer other
// instructions per MFMA is the only lever.  Same arithmetic per output in the same order: bit-identical to NB = 1.
template <int H_, int NB>
__global__ void __launch_bounds__(WN, NB == 2 ? 2 : 3)
wino_conv_kernel(const vv_conv_params p, const int NT, const int NN, const int total, const int nper) {
  using G_ = WGeo<H_>;
  constexpr int TPI = G_::TPI, TPW = G_::TPW, NI = G_::NI, PARTS = G_::PARTS, HH = G_::HH, HW = G_::HW, HWH = HW / 2;
  constexpr int CK = 8, S = CK + 4, S4 = S / 4, Q = CK / 4;
  constexpr int A4 = NI * HH * HW * S4;
  constexpr int NITEMS = NI * HH * HW * Q;
  constexpr int NIT = (NITEMS + WN - 1) / WN;
  constexpr int EX4 = 4 * 16 * 64 / 2;                     // epilogue exchange: [wave][16 regs][64 lanes] float2
  constexpr int L4 = A4 > EX4 ? A4 : EX4;
  static_assert(WN % Q == 0, "one channel group per thread");
  __shared__ float4 lds4[L4 + 64];                        // + [2][4 waves][32] BatchNorm partials
  float* lds = reinterpret_cast<float*>(lds4);
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int nn = w % NN; w /= NN;      // the N tiles of one pixel tile are neighbours in launch order: they share the halo in L2
  const int pt = w % NT;
  const int g = w / NT;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img0 = (pt / PARTS) * NI, part = pt % PARTS;
  const int y0 = part * (2 * G_::TROWS) - 1;               // conv-input row of halo row 0 (column origin is -1)
  const VVSrc s = vv_make_src(p, g, H_, H_);
  const int Cout = p.Cout, CinP = p.CinP, KQ = CinP >> 3;
  const int co0 = nn * 32 * NB;
  const float* __restrict__ wg = p.w + (int64_t)g * p.w_gstride;

  // ---- halo staging set-up
  float4 r[NIT];
  int slot[NIT];
  unsigned valid = 0;
  unsigned voff[NIT];
  int pixv[NIT];
  const int tile = (img0 * H_ + y0) * H_ - 1;
  const int q4 = (tid % Q) * 4;
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = tid + k * WN;
    const int q = it % Q, hp = it / Q;
    const int hx = hp % HW, t = hp / HW;
    const int hy = t % HH, im = t / HH;
    const bool inr = NITEMS % WN == 0 || it < NITEMS;
    const int y = y0 + hy;
    const bool ok = inr && (unsigned)(hx - 1) < (unsigned)H_ && (unsigned)y < (unsigned)H_ && (img0 + im) < s.B;
    valid |= ok ? (1u << k) : 0u;
    pixv[k] = (im * H_ + hy) * H_ + hx;
    slot[k] = inr ? ((im * HH + hy) * HW + (hx & 1) * HWH + (hx >> 1)) * S4 + q : -1;
  }
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned bvo = (unsigned)(half * Cout + co0 + l31) * 16u;                // this lane inside a [2][Cout] float4 slab
  const int bnu = KQ * 2 * Cout * 16;                                            // bytes between nu slabs
  const int bxi = xi * 4 * bnu;
  float4 sa, sb;
  bool act = false;
  __amdgpu_buffer_rsrc_t rs;
  int cur_second = -1, soff0 = 0;
  auto set_source = [&](const bool second) {
    const float* base = second ? s.p1 + s.co1 : s.p0 + s.co0;
    const int cs = second ? s.cs1 : s.cs0;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
    soff0 = second ? -s.csplit * 4 : 0;
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      voff[k] = ((valid >> k) & 1u) ? (unsigned)((tile + pixv[k]) * cs + q4) * 4u : 0x80000000u;
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
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (NITEMS % WN == 0 || k < NIT - 1 || slot[k] >= 0) {
        float4 v = r[k];
        if (act && ((valid >> k) & 1u)) v = vv_act4(v, sa, sb);
        lds4[slot[k]] = v;
      }
  };
  auto load_u = [&](const int c0, const int n, const int nb) -> v4f {
    return __builtin_amdgcn_raw_buffer_load_b128(rsW, bvo + nb * 512u, bxi + n * bnu + (c0 >> 3) * 2 * Cout * 16, 0);
  };

  // ---- this lane's tile and its patch origin in LDS
  const int tim = l31 / TPW, trem = l31 % TPW;
  const int tyl = trem / TPI, tx = trem % TPI;
  const int pbase = ((tim * HH + 2 * tyl) * HW + tx) * S4 + half;
  auto patch = [&](const int a, const int b) -> v4f {
    return ldsA[pbase + (a * HW + (b & 1) * HWH + (b >> 1)) * S4];
  };
  // B^T rows: xi 0: d0-d2   1: d1+d2   2: d2-d1   3: d1-d3   ->   R = d[a1] + sg * d[a2]
  const int a1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), a2 = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
  const float sg = xi == 1 ? 1.f : -1.f;

  v16f acc[NB][4];
  v4f u[NB][4];
  // Order of the memory instructions (loads return in order, one counter): the halo loads of the NEXT chunk are issued at the
  // start of a chunk, before this chunk's tap loads; each tap load goes out right after the four MFMAs that read the registers
  // it overwrites, one chunk ahead of its use.  sched_barrier keeps the compiler from sinking the loads to their first use
  // (left alone it does: one L2 round trip per chunk with nothing else in flight); VALU / SALU / LDS instructions may cross.
  const int klast = CinP - CK;
  auto compute = [&](const int c0, const auto first, const auto more) {
    const int kn = c0 + CK;
    const int knext = kn < klast ? kn : klast;             // the last chunk re-reads its own taps (unused)
    // (compile-time: a run-time branch around the loads makes the wait-count pass assume the shorter queue on both sides)
    if constexpr (decltype(more)::value) issue(kn);
    v4f R[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const v4f d1 = patch(a1, b), d2 = patch(a2, b);
      R[b] = d1 + sg * d2;
    }
    v4f V[4];
    V[0] = R[0] - R[2];
    V[1] = R[1] + R[2];
    V[2] = R[2] - R[1];
    V[3] = R[1] - R[3];
    __builtin_amdgcn_sched_barrier(SB_MASK);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if constexpr (decltype(first)::value) {   // the accumulators start from the instruction's inline-constant 0
          const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[nb][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].x, u[nb][n].x, z, 0, 0, 0);
        } else {
          acc[nb][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].x, u[nb][n].x, acc[nb][n], 0, 0, 0);
        }
        acc[nb][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].y, u[nb][n].y, acc[nb][n], 0, 0, 0);
        acc[nb][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].z, u[nb][n].z, acc[nb][n], 0, 0, 0);
        acc[nb][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].w, u[nb][n].w, acc[nb][n], 0, 0, 0);
        u[nb][n] = load_u(knext, n, nb);
        __builtin_amdgcn_sched_barrier(SB_MASK);
      }
    }
  };

  issue(0);
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) u[nb][n] = load_u(0, n, nb);
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
  // (one N tile after the other through the same exchange region)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int cob = co0 + nb * 32;
    __syncthreads();                    // all MFMA-phase LDS reads (nb = 0) / the previous N tile's exchange and sums (nb = 1) are done
    v2f* ex2 = reinterpret_cast<v2f*>(lds) + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const v2f t = {acc[nb][0][i] + acc[nb][1][i] + acc[nb][2][i], acc[nb][1][i] - acc[nb][2][i] - acc[nb][3][i]};
      ex2[(xi * 16 + i) * 64] = t;
    }
    constexpr int LP = TPI > 4 ? 8 : (TPI == 4 ? 2 * H_ : H_ * H_);        // pixels between the two lane halves (tile + 4)
    // Fused first pass of the consumer's BatchNorm backward (vv_conv_params.bn_partial): this wave's 16 values of z go out now and
    // land while the four xi rows meet in LDS.
    const bool bnf = p.bn_partial != nullptr;
    float zq[4][4];
    float bna = 0.f, bnb = 0.f, bni = 0.f, bnm = 0.f;
    bool jok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t2 = xi * 8 + j;
      jok[j] = img0 + t2 / TPW + (TPW == 4 ? half : 0) < p.B;
    }
    if (bnf) {
      const int64_t bo = (int64_t)g * p.bn_gstride + cob + l31;
      bna = p.bn_a[bo]; bnb = p.bn_b[bo]; bni = p.bn_invstd[bo];
      bnm = -p.bn_mean[bo] * bni;                                            // xhat = z invstd - mean invstd
      const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.bn_z + (int64_t)g * p.bn_z_gstride), 0, 0x7FFFFFFF, 0x00020000);
      const int vz = (half * LP * Cout + cob + l31) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t2 = xi * 8 + j;
        const int im = t2 / TPW, rem = t2 % TPW;
        const int oy = 2 * (part * G_::TROWS + rem / TPI), ox = 2 * (rem % TPI);
        const int so = (((img0 + im) * H_ + oy) * H_ + ox) * Cout * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          zq[j][q] = jok[j] ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsZ, vz, so + ((q >> 1) * H_ + (q & 1)) * Cout * 4, 0))
                            : 0.f;
      }
    }
    __syncthreads();
    const float bias = p.bias ? p.bias[(int64_t)g * p.bias_gstride + cob + l31] : 0.f;
    const float lo = (p.pad0 & VV_CONV_RELU) ? 0.f : -__builtin_inff();     // VV_CONV_RELU: BatchNorm folded into the filter (eval)
    const int ocs = p.out.cstride;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
        p.out.ptr + (int64_t)g * p.out.gstride + p.out.coff, 0, 0x7FFFFFFF, 0x00020000);
    const int vo = (half * LP * ocs + cob + l31) * 4;
    const v2f* exw = ex2 + xi * 4 * 64;
    v2f s12 = {0.f, 0.f}, q12 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t2 = xi * 8 + j;                                   // wave-uniform part of the tile index
      const int im = t2 / TPW, rem = t2 % TPW;
      const int oy = 2 * (part * G_::TROWS + rem / TPI), ox = 2 * (rem % TPI);
      if (jok[j]) {
        const v2f t0 = exw[(0 * 16 + j) * 64], t1 = exw[(1 * 16 + j) * 64], t2v = exw[(2 * 16 + j) * 64], t3 = exw[(3 * 16 + j) * 64];
        v2f ya = t0 + t1 + t2v + bias, yb = t1 - t2v - t3 + bias;
        // ReLU of the folded eval path (lo = 0; train mode: lo = -inf, a no-op).  Compare + select, not v_max: v_max_f32 returns the
        // non-NaN operand, which would turn a diverged model's NaN into 0 / -inf where torch's ReLU propagates it
        ya[0] = ya[0] < lo ? lo : ya[0]; ya[1] = ya[1] < lo ? lo : ya[1];
        yb[0] = yb[0] < lo ? lo : yb[0]; yb[1] = yb[1] < lo ? lo : yb[1];
        const int so = (((img0 + im) * H_ + oy) * H_ + ox) * ocs * 4;
        const float a0 = ya[0], a1v = ya[1], b0 = yb[0], b1 = yb[1];
        {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a0), rsO, vo, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a1v), rsO, vo, so + ocs * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(b0), rsO, vo, so + H_ * ocs * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(b1), rsO, vo, so + (H_ + 1) * ocs * 4, 0);
        }
        if (bnf) {
          // dz = dA [a z + b > 0];  partial sums of dz and dz * xhat  (bn_bwd_reduce_kernel<.., 0>, fused)
          const v2f za = {zq[j][0], zq[j][1]}, zb = {zq[j][2], zq[j][3]};
          const v2f pa = bna * za + bnb, pb = bna * zb + bnb;
          const v2f da = {pa.x > 0.f ? ya.x : 0.f, pa.y > 0.f ? ya.y : 0.f}, db = {pb.x > 0.f ? yb.x : 0.f, pb.y > 0.f ? yb.y : 0.f};
          s12 += da + db;
          q12 = __builtin_elementwise_fma(da, bni * za + bnm, q12);
          q12 = __builtin_elementwise_fma(db, bni * zb + bnm, q12);
        } else {
          s12 += ya + yb;
          q12 = __builtin_elementwise_fma(ya, ya, q12);
          q12 = __builtin_elementwise_fma(yb, yb, q12);
        }
      }
    }
    float* const sout = bnf ? p.bn_partial : p.stats;
    if (sout) {
      float s1 = s12.x + s12.y, s2 = q12.x + q12.y;
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      float* sp = lds + L4 * 4;
      if (half == 0) {
        sp[xi * 32 + l31] = s1;
        sp[(4 + xi) * 32 + l31] = s2;
      }
      __syncthreads();
      if (tid < 32) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          t1 += sp[k * 32 + tid];
          t2 += sp[(4 + k) * 32 + tid];
        }
        float* st = sout + ((int64_t)(g * NT + pt) * 2) * Cout + cob + tid;
        st[0] = t1;
        st[Cout] = t2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// wino_ring_kernel<KQ, BNF>: the same convolution on the 32x32 level for layers with at most 32 (padded) input channels (KQ = CinP
// / 8 chunks: conv0, conv1 / 13, their data gradients and the data gradient of the concat layer -- 6 of the 7 launches of the
// level), built around the memory system instead of around the workgroup.
//
// Why: wino_conv_kernel<32> runs these layers at 0.40 of the matrix peak with the matrix pipe half idle.  A workgroup lives ~8 us
// for 1.7 us of MFMA work: 2 us until its first halo chunk is back from HBM, up to 1 us more at every chunk boundary (a chunk's 16
// MFMAs are shorter than the round trip of the next chunk's loads), an epilogue that loads nothing.  Three workgroups per CU (158
// registers) then keep ~20 KB per CU in flight = 2.5 TB/s by Little's law -- exactly what the launch moves.  Deeper prefetch inside
// a workgroup does not help (round 4: all chunks requested at once, same time): the average in flight over the workgroup's LIFE is
// what counts, and the in-order load counter ties the halo prefetch to the filter-tap loads of the next chunk.
//
// Here a workgroup is PERSISTENT (two per CU, a contiguous run of pixel tiles each) and
//   * the transformed filter taps of its (UNet, N tile) live in registers for the whole run (KQ x 16 per lane: possible because
//     K <= 32) -- the chunk loop has NO register-destination loads, so the load counter counts halo traffic only;
//   * halo chunks go global -> LDS by DMA (buffer_load_dwordx4 ... lds, no staging registers) into a ring of NBUF = D + 1 chunk
//     buffers, D chunks = one or two whole tiles AHEAD of the chunk being multiplied, across tile boundaries and across the
//     epilogue: the HBM round trip of a tile overlaps the MFMAs and the epilogue of the tile before it.  Counted s_waitcnt vmcnt +
//     raw s_barrier (a __syncthreads() would drain the ring); the stores of an epilogue sit in the same in-order counter and are
//     counted (STORES_MIN), so a chunk's wait never waits for them;
//   * the producing layer's BatchNorm + ReLU is applied in place in LDS by the lane that transferred the 16 bytes (own vmcnt, no
//     barrier in between); zero padding = DMA lanes whose offset is out of range (zeros) and that the activation pass skips;
//   * ONE barrier per chunk (wino_conv_kernel: two): a ring slot is refilled right behind the barrier that proves its last
//     reader has passed.
// LDS image of a chunk: two planes (channels 0-3 / 4-7 of the 8-channel chunk = the two lane halves of an A operand) of
// [6 halo rows][column parity][17] float4, no padding: the 16 lanes of a tile row read 16 consecutive float4 (conflict-free), and
// the image is lane-linear as the DMA needs it (slot s = wave * 128 + k * 64 + lane; 408 of 512 slots used).
// Same arithmetic in the same order as wino_conv_kernel<32> (chunk -> nu -> 4 MFMAs, same epilogue): bit-identical outputs and
// BatchNorm partial sums (tests/test_gpu_unet.py::test_wino_ring_bitwise_equal).
template <int KQ, bool BNF, bool RELU>
__global__ void __launch_bounds__(WN, 2)
wino_ring_kernel(const vv_conv_params p, const int NT, const int NN, const int total, const int ipw) {
  constexpr int H_ = 32, TPI = 16, PARTS = 8, HWH = 17;
  constexpr int D = VV_RING_D;                             // prefetch distance in chunks (whole tiles: D % KQ == 0)
  constexpr int E = D / KQ;                                // ... = tiles ahead
  constexpr int NBUF = D + 1;
  constexpr int CH4 = 512;                                 // float4 slots per ring buffer (408 used)
  constexpr int PLANE = 6 * 2 * HWH;                       // 204 float4 per channel-quad plane
  constexpr int NUSED = 2 * PLANE;                         // 408
  constexpr int EX4 = 4 * 16 * 64 / 2;                     // epilogue exchange [wave][16][64] float2
  constexpr int RING4 = NBUF * CH4;
  constexpr int SP4 = RING4 + EX4;                         // [2][4][64] floats: BatchNorm partials of the four waves' lanes
  constexpr int AB4 = SP4 + 128;                           // [2][KQ * 2] float4: a, b of the producing layer's BatchNorm
  // COMPILER DEPENDENCY: the counted waits below assume the epilogue's output stores stay >= STORES_MIN separate VMEM instructions per
  // wave.  tests/test_build_invariants.py checks that on the emitted assembly at every build (no GPU needed); VV_WINO_RING=0 /
  // VV_CONV_NO_RING routes the same launches to the per-tile kernel, whose waits are the compiler's own.
  constexpr int STORES_MIN = 16;                           // VMEM instructions every wave issues in every epilogue, at least
  constexpr int VMW = 2 * (D - 1) + STORES_MIN * E;        // allowed outstanding when chunk j must have landed (see above)
  static_assert(D % KQ == 0 && VMW <= 63, "ring geometry");
  __shared__ float4 lds4[AB4 + 4 * KQ];                    // the ONLY __shared__ object (73.6 KB: two workgroups per CU)
  float* lds = reinterpret_cast<float*>(lds4);
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds4;

  const int w_begin = blockIdx.x * ipw;
  const int w_end = w_begin + ipw < total ? w_begin + ipw : total;
  if (w_begin >= w_end) return;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cout = p.Cout;
  const bool act_mode = p.in_mode == VV_IN_ACT;
  const int cs = p.src0.cstride;

  // ---- this lane's two DMA items per chunk (k = 0, 1): slot s -> (channel quad q, halo row hy, column parity, column index)
  int rel[2];                  // byte offset inside the halo tile, or out of range
  int yflag[2];                // 1: top halo row (outside the image for part 0), 2: bottom halo row (outside for the last part)
  bool xok[2];
  int qk[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int sl = xi * 128 + k * 64 + lane;
    const int q = sl >= PLANE ? 1 : 0, r = sl - q * PLANE;
    const int hy = r / (2 * HWH), r2 = r % (2 * HWH);
    const int par = r2 >= HWH ? 1 : 0, cx = r2 - par * HWH;
    const int x = 2 * cx + par - 1;
    xok[k] = sl < NUSED && (unsigned)x < (unsigned)H_;
    rel[k] = ((hy * H_ + x) * cs + 4 * q) * 4;
    yflag[k] = (hy == 0 ? 1 : 0) | (hy == 5 ? 2 : 0);
    qk[k] = sl < NUSED ? q : 0;
  }

  // ---- cursors.  Work item w -> (g, nn, pt), pt fastest: a workgroup's run stays inside one (UNet, N tile) as long as possible
  auto decode = [&](const int w, int& g, int& nn, int& pt) {
    pt = w % NT;
    const int t = w / NT;
    nn = t % NN;
    g = t / NN;
  };
  int gc, nc, ptc;             // compute cursor
  decode(w_begin, gc, nc, ptc);
  int gd = gc, nd = nc, ptd = ptc, wd = w_begin;         // DMA cursor (E tiles ahead in steady state)
  auto advance = [&](int& g, int& nn, int& pt) {
    if (++pt == NT) {
      pt = 0;
      if (++nn == NN) { nn = 0; ++g; }
    }
  };
  unsigned voff[2] = {0x80000000u, 0x80000000u};
  __amdgpu_buffer_rsrc_t rsS;
  auto dma_item = [&]() {      // per-item part of the DMA addresses for the tile under the DMA cursor (or "nothing": past the run)
    const bool live = wd < w_end;
    const int img = ptd >> 3, part = ptd & 7;
    const int tileoff = ((img * H_ + part * 4 - 1) * H_) * cs * 4;
    const int ym = (part == 0 ? 1 : 0) | (part == PARTS - 1 ? 2 : 0);
    rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src0.ptr + (int64_t)(live ? gd : gc) * p.src0.gstride + p.src0.coff), 0,
                                            0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      voff[k] = (live && xok[k] && !(yflag[k] & ym)) ? (unsigned)(tileoff + rel[k]) : 0x80000000u;
  };
  int slotd = 0;               // ring slot the next DMA pair fills
  auto dma_chunk = [&](const int c) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned dst = lds_base + (unsigned)((slotd * CH4 + xi * 128 + k * 64) * 16);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   ::"s"(dst), "v"(voff[k]), "s"(rsS), "s"(c * 32) : "memory", "m0");
    }
    slotd = slotd + 1 == NBUF ? 0 : slotd + 1;
  };

  // ---- prologue: the first E tiles' chunks in flight
#pragma unroll
  for (int e = 0; e < E; ++e) {
    dma_item();
#pragma unroll
    for (int c = 0; c < KQ; ++c) dma_chunk(c);
    advance(gd, nd, ptd);
    ++wd;
  }

  // ---- this lane's tile and patch origin inside a ring buffer
  const int tyl = l31 >> 4, tx = l31 & 15;
  const int pbase = half * PLANE + (2 * tyl) * 2 * HWH + tx;
  const int a1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), a2 = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
  const float sg = xi == 1 ? 1.f : -1.f;
  const int po1 = pbase + a1 * 2 * HWH, po2 = pbase + a2 * 2 * HWH;

  v4f u[KQ][4];
  float bias = 0.f;
  float bna = 0.f, bnb = 0.f, bni = 0.f, bnm = 0.f;
  int g_have = -1, n_have = -1;
  const int ocs = p.out.cstride;
  constexpr int LP = 8;                                    // pixels between the two lane halves (tile + 4)
  const int bnu = KQ * 2 * Cout * 16;                      // bytes between nu slabs of the packed panel
  int slotc = 0;
  v2f* ex2 = reinterpret_cast<v2f*>(lds4 + RING4) + lane;
  float* sp = lds + SP4 * 4;                              // [sum | sum of squares][wave][64 lanes]
  float* const sout = BNF ? p.bn_partial : p.stats;
  bool pend = false;                                      // a tile's column sums wait in sp for their reduction
  int64_t pend_off = 0;
  auto flush_stats = [&]() {                              // wave w reduces and stores channels 8w .. 8w+7
    if (lane < 8) {
      const int ch = xi * 8 + lane;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t1 += sp[k * 64 + ch] + sp[k * 64 + 32 + ch];
        t2 += sp[(4 + k) * 64 + ch] + sp[(4 + k) * 64 + 32 + ch];
      }
      float* st = sout + pend_off + ch;
      st[0] = t1;
      st[Cout] = t2;
    }
    pend = false;
  };

  // ---- the chunk pipeline.  A wave's own latencies (DMA issue, the wait for the next chunk, its activation, the barrier, the
  //      patch reads) sit INSIDE the 16 MFMAs of the chunk before: k steps x, y | DMA issue for chunk j + D, wait for chunk j + 1,
  //      activate this wave's pieces of it | k step z | barrier, patch reads of chunk j + 1 | k step w | input transform of chunk
  //      j + 1 (VALU: it cannot run under fp32 MFMAs of the same SIMD, so it is not interleaved).  Profiled before this order
  //      (s_memtime per phase, 32 -> 32): 14.6 k cycles per tile and wave, 3.9 k of them issuing MFMAs, the rest a serial chain
  //      of those latencies.
  auto land = [&](const int cn, const int ym, const int slot, const bool same_tile) {   // this wave's pieces of a chunk: landed, activated
    // in-order counter: everything issued after the chunk's DMA pair may stay outstanding -- the D - 1 younger pairs and the
    // stores of the epilogues in between (E of them; E - 1 when the chunk is the first of the NEXT tile, whose pair went out D
    // chunks ago = after the last epilogue but E - 1)
    if (same_tile)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW - STORES_MIN) : "memory");
    if (act_mode) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (xok[k] && !(yflag[k] & ym)) {
          const int sl = slot * CH4 + xi * 128 + k * 64 + lane;
          const float4 a4 = lds4[AB4 + cn * 2 + qk[k]], b4 = lds4[AB4 + 2 * KQ + cn * 2 + qk[k]];
          lds4[sl] = vv_act4(lds4[sl], a4, b4);
        }
    }
  };
  v4f P[8];
  auto fetch = [&](const int slot) {
    const int rb = slot * CH4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int o = (b & 1) * HWH + (b >> 1);
      P[2 * b] = ldsA[rb + po1 + o];
      P[2 * b + 1] = ldsA[rb + po2 + o];
    }
  };
  v4f V[4];
  auto transform = [&]() {
    v4f R[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) R[b] = P[2 * b] + sg * P[2 * b + 1];
    V[0] = R[0] - R[2];
    V[1] = R[1] + R[2];
    V[2] = R[2] - R[1];
    V[3] = R[1] - R[3];
  };
  bool restart = true;         // the pipeline has no chunk in its registers: the first tile, and the first tile of another UNet

  for (int w = w_begin; w < w_end; ++w) {
    const int co0 = nc * 32;
    const int img = ptc >> 3, part = ptc & 7;
    const int ymc = (part == 0 ? 1 : 0) | (part == PARTS - 1 ? 2 : 0);
    if (gc != g_have || nc != n_have) {
      // ---- a new (UNet, N tile): filter taps -> registers, bias / BatchNorm scalars, activation table -> LDS.  Rare (a run of
      //      tiles shares them); these register loads drain the DMA ring once (in-order counter), the counted waits below stay valid
      const float* __restrict__ wg = p.w + (int64_t)gc * p.w_gstride;
      const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, 0x7FFFFFFF, 0x00020000);
      const unsigned bvo = (unsigned)(half * Cout + co0 + l31) * 16u;
#pragma unroll
      for (int c = 0; c < KQ; ++c)
#pragma unroll
        for (int n = 0; n < 4; ++n) u[c][n] = __builtin_amdgcn_raw_buffer_load_b128(rsW, bvo, (xi * 4 + n) * bnu + c * 2 * Cout * 16, 0);
      bias = p.bias ? p.bias[(int64_t)gc * p.bias_gstride + co0 + l31] : 0.f;
      if constexpr (BNF) {
        const int64_t bo = (int64_t)gc * p.bn_gstride + co0 + l31;
        bna = p.bn_a[bo]; bnb = p.bn_b[bo]; bni = p.bn_invstd[bo];
        bnm = -p.bn_mean[bo] * bni;
      }
      if (act_mode && gc != g_have) {
        if (tid < 4 * KQ) {
          const float* src = (tid < 2 * KQ ? p.a : p.b) + (int64_t)gc * p.ab_gstride + (tid % (2 * KQ)) * 4;
          lds4[AB4 + tid] = *reinterpret_cast<const float4*>(src);
        }
      }
      __builtin_amdgcn_s_waitcnt(0);             // (the builtin, not asm: the compiler's own load scoreboard must see it, or it waits
      vv_lds_barrier();                          //  vmcnt(0) in front of the first MFMA of EVERY tile and drains the ring there)
      g_have = gc; n_have = nc;
    }
    if (restart) {             // chunk 0 of this tile: landed (everything is: the tap loads above drained the counter), activated, read
      land(0, ymc, slotc, false);
      vv_lds_barrier();
      fetch(slotc);
      transform();
      restart = false;
    }
    // the tile after this one (its first chunk is prepared under this tile's last MFMAs, with THIS UNet's activation table)
    int gn = gc, nn_ = nc, ptn = ptc;
    advance(gn, nn_, ptn);
    const bool cross = gn != gc;                 // ... unless it belongs to another UNet: then the pipeline restarts there
    const int ymn = ((ptn & 7) == 0 ? 1 : 0) | ((ptn & 7) == PARTS - 1 ? 2 : 0);

    v16f acc[4];
#pragma unroll
    for (int c = 0; c < KQ; ++c) {
      const bool last = c == KQ - 1;
      const int slotn = slotc + 1 == NBUF ? 0 : slotc + 1;
      auto mfma_k = [&](const int k) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if (c == 0 && k == 0) {
            const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n][k], u[c][n][k], z, 0, 0, 0);
          } else {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n][k], u[c][n][k], acc[n], 0, 0, 0);
          }
        }
      };
      // (the four GEMMs' MFMAs interleaved, k step outermost: consecutive matrix instructions write DIFFERENT accumulators, each
      //  accumulator sees its k steps in the order x, y, z, w -- bit-identical to the per-tile kernel's nu-major issue)
      __builtin_amdgcn_sched_barrier(0);
      mfma_k(0);
      mfma_k(1);
      __builtin_amdgcn_sched_barrier(0);
      // every wave is past the barrier of chunk j: the slot of chunk j - 1 takes chunk j + D (same chunk index, E tiles ahead)
      if (c == 0) dma_item();
      dma_chunk(c);
      if (last) { advance(gd, nd, ptd); ++wd; }
      if (!(last && cross)) land(last ? 0 : c + 1, last ? ymn : ymc, slotn, !last);
      __builtin_amdgcn_sched_barrier(0);
      mfma_k(2);
      __builtin_amdgcn_sched_barrier(0);
      vv_lds_barrier();                          // chunk j + 1 is complete in LDS; the column sums of the tile before are too
      if (c == 0 && pend) flush_stats();
      if (!(last && cross)) fetch(slotn);
      __builtin_amdgcn_sched_barrier(0);
      mfma_k(3);
      __builtin_amdgcn_sched_barrier(0);
      if (!(last && cross)) transform();
      slotc = slotn;
    }
    if (cross) restart = true;

    // ---- epilogue (wino_conv_kernel's, on its own LDS region: the ring keeps filling underneath)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const v2f t = {acc[0][i] + acc[1][i] + acc[2][i], acc[1][i] - acc[2][i] - acc[3][i]};
      ex2[(xi * 16 + i) * 64] = t;
    }
    float zq[4][4];
    if constexpr (BNF) {
      const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.bn_z + (int64_t)gc * p.bn_z_gstride), 0, 0x7FFFFFFF, 0x00020000);
      const int vz = (half * LP * Cout + co0 + l31) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t2 = xi * 8 + j;
        const int oy = 2 * (part * 2 + t2 / TPI), ox = 2 * (t2 % TPI);
        const int so = ((img * H_ + oy) * H_ + ox) * Cout * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          zq[j][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsZ, vz, so + ((q >> 1) * H_ + (q & 1)) * Cout * 4, 0));
      }
    }
    vv_lds_barrier();
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
        p.out.ptr + (int64_t)gc * p.out.gstride + p.out.coff, 0, 0x7FFFFFFF, 0x00020000);
    const int vo = (half * LP * ocs + co0 + l31) * 4;
    const v2f* exw = ex2 + xi * 4 * 64;
    v2f s12 = {0.f, 0.f}, q12 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t2 = xi * 8 + j;
      const int oy = 2 * (part * 2 + t2 / TPI), ox = 2 * (t2 % TPI);
      const v2f t0 = exw[(0 * 16 + j) * 64], t1 = exw[(1 * 16 + j) * 64], t2v = exw[(2 * 16 + j) * 64], t3 = exw[(3 * 16 + j) * 64];
      v2f ya = t0 + t1 + t2v + bias, yb = t1 - t2v - t3 + bias;
      if constexpr (RELU) {        // VV_CONV_RELU (folded eval model); compare + select so that a NaN stays a NaN, like torch's ReLU
        ya[0] = ya[0] < 0.f ? 0.f : ya[0]; ya[1] = ya[1] < 0.f ? 0.f : ya[1];
        yb[0] = yb[0] < 0.f ? 0.f : yb[0]; yb[1] = yb[1] < 0.f ? 0.f : yb[1];
      }
      const int so = ((img * H_ + oy) * H_ + ox) * ocs * 4;
      {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ya[0]), rsO, vo, so, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ya[1]), rsO, vo, so + ocs * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yb[0]), rsO, vo, so + H_ * ocs * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yb[1]), rsO, vo, so + (H_ + 1) * ocs * 4, 0);
      }
      if constexpr (BNF) {
        const v2f za = {zq[j][0], zq[j][1]}, zb = {zq[j][2], zq[j][3]};
        const v2f pa = bna * za + bnb, pb = bna * zb + bnb;
        const v2f da = {pa.x > 0.f ? ya.x : 0.f, pa.y > 0.f ? ya.y : 0.f}, db = {pb.x > 0.f ? yb.x : 0.f, pb.y > 0.f ? yb.y : 0.f};
        s12 += da + db;
        q12 = __builtin_elementwise_fma(da, bni * za + bnm, q12);
        q12 = __builtin_elementwise_fma(db, bni * zb + bnm, q12);
      } else {
        s12 += ya + yb;
        q12 = __builtin_elementwise_fma(ya, ya, q12);
        q12 = __builtin_elementwise_fma(yb, yb, q12);
      }
    }
    if (sout) {
      // this lane's column sums go to LDS as they are; the sum over the two lane halves and the four waves (in the per-tile
      // kernel's order: (half 0 + half 1) per wave, then waves 0..3) and the store happen behind the NEXT tile's first chunk
      // barrier, which proves that every wave has written them -- no barrier of its own, no cross-half shuffles
      sp[xi * 64 + lane] = s12.x + s12.y;
      sp[(4 + xi) * 64 + lane] = q12.x + q12.y;
      pend_off = ((int64_t)(gc * NT + ptc) * 2) * Cout + co0;
      pend = true;
    }
    advance(gc, nc, ptc);
  }
  if (pend) {
    vv_lds_barrier();
    flush_stats();
  }
  // the tail's dummy DMAs write zeros into ring slots: they must have landed before this workgroup's LDS is handed on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// wino_ring_kernel: two persistent workgroups per CU, a contiguous run of work items each
template <int KQ>
int launch_wino_ring(const vv_conv_params* p, hipStream_t st) {
  const int NT = p->B * 8;
  const int NN = p->Cout / 32;
  const int total = p->G * NN * NT;
  const int slots = 2 * vv_num_cus();
  const int ipw = (total + slots - 1) / slots;
  const int nwg = (total + ipw - 1) / ipw;
  if (p->bn_partial)          // (data-gradient launches: never with the eval path's ReLU)
    VV_LAUNCH((wino_ring_kernel<KQ, true, false>), dim3(nwg), dim3(WN), 0, st, *p, NT, NN, total, ipw);
  else if (p->pad0 & VV_CONV_RELU)
    VV_LAUNCH((wino_ring_kernel<KQ, false, true>), dim3(nwg), dim3(WN), 0, st, *p, NT, NN, total, ipw);
  else
    VV_LAUNCH((wino_ring_kernel<KQ, false, false>), dim3(nwg), dim3(WN), 0, st, *p, NT, NN, total, ipw);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

// the launches wino_ring_kernel takes: 32x32 level, K <= 32, one plain or activated source, enough tiles for runs of >= 4 per workgroup
inline bool vv_wino_ring_ok(const vv_conv_params* p) {
  if (p->H != 32 || (p->CinP != 16 && p->CinP != 32) || (p->pad0 & VV_CONV_NO_RING)) return false;
  if (p->in_mode != VV_IN_PLAIN && p->in_mode != VV_IN_ACT) return false;
  if (p->bn_partial && (p->pad0 & VV_CONV_RELU)) return false;
  return (int64_t)p->G * (p->Cout / 32) * p->B * 8 >= VV_RING_MIN;
}

template <int H_>
int launch_wino(const vv_conv_params* p, hipStream_t st) {
  using G_ = WGeo<H_>;
  const int NT = ((p->B + G_::NI - 1) / G_::NI) * G_::PARTS;
  // two N tiles per workgroup (two workgroups per CU) where that leaves whole rounds of 512 workgroups: at least two, the last one
  // at least 90 % full; VV_WINO_NB=1 in the environment keeps one N tile everywhere (A/B switch, read once)
  static const bool nb1 = [] { const char* e = getenv("VV_WINO_NB"); return e && e[0] == '1'; }();
  bool two = !nb1 && p->Cout % 64 == 0;
  if (two) {
    const int64_t wgs = (int64_t)p->G * (p->Cout / 64) * NT;
    const int64_t rounds = (wgs + 511) / 512;
    two = wgs >= 1024 && wgs * 10 >= rounds * 512 * 9;
  }
  const int NN = p->Cout / (two ? 64 : 32);
  const int total = p->G * NN * NT;
  const int nper = (total + 7) / 8;
  if (two)
    VV_LAUNCH((wino_conv_kernel<H_, 2>), dim3(nper * 8), dim3(WN), 0, st, *p, NT, NN, total, nper);
  else
    VV_LAUNCH((wino_conv_kernel<H_, 1>), dim3(nper * 8), dim3(WN), 0, st, *p, NT, NN, total, nper);
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
  if (p->out1.ptr) return VV_ERR_UNSUPPORTED;                          // second output view: bf16-output launches of vv_conv_mfma
  if (p->bn_partial && (p->stats || !p->bn_z || !p->bn_a || !p->bn_b || !p->bn_mean || !p->bn_invstd)) return VV_ERR_BAD_ARG;
  {
    // the epilogue addresses one UNet's output (and z) with 32-bit byte offsets
    const int64_t px = (int64_t)p->B * p->H * p->W * 4;
    const int64_t cs = p->out.cstride > p->Cout ? p->out.cstride : p->Cout;
    if (px * cs >= (1ll << 31)) return VV_ERR_UNSUPPORTED;
  }
  if (p->in_mode == VV_IN_POOL || p->in_mode == VV_IN_CUBE)
    return VV_ERR_UNSUPPORTED;    // feed the materialised tensor (vv_pool_act / vv_cube_erase) as VV_IN_PLAIN
  hipStream_t st = (hipStream_t)stream;
  if (vv_wino_ring_ok(p)) return p->CinP == 16 ? launch_wino_ring<2>(p, st) : launch_wino_ring<4>(p, st);
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
