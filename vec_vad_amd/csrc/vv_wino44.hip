// Winograd F(4x4, 3x3) convolution on the matrix cores (gfx950, v_mfma_f32_32x32x2_f32) for the UNet bank's 3x3 / stride 1 /
// pad 1 layers (model/unet.py:10,13) -- forward and data-gradient, round 5.
//
//   Y = A^T [ (G g G^T) . (B^T d B) ] A     per 6x6 input patch d -> 4x4 outputs, summed over input channels
//
// 36 element-wise products per 16 output pixels = 2.25 matrix-core multiply-adds per output and channel pair, against 4 for
// F(2x2,3x3) (vv_wino.hip) and 9 for the direct form: 1.78x fewer MFMA cycles than the kernel this one replaces on the layers that
// are bound by the matrix pipe (>= 64 input channels).  All in fp32.  The transforms multiply by up to 8 (A) / 5 (B), so the result
// is 1 - 3e-6 of the tensor's maximum from the direct convolution with the interpolation points used here (the textbook points: 4 -
// 14e-6; F(2x2): a few 1e-7); what the textbook form does to the 1e-3 bars of a training trajectory was measured before this kernel
// was written (profiles/r05_wino44_numerics.txt, tests/numerics_wino44.py).
//
// A group of 384 threads = 6 waves works on 32 tiles (512 output pixels) x 32 output channels of one UNet; wave = xi (0..5), its six
// GEMMs (nu = 0..5) live in 96 accumulator registers, <= 168 registers per lane = three waves per SIMD.  A WORKGROUP is two such
// groups on neighbouring pixel tiles (768 threads, separate LDS images, shared barriers): six waves land 2 + 2 + 1 + 1 on the four
// SIMDs and a second six-wave workgroup did not fit beside them (measured: half the occupancy), twelve land 3 + 3 + 3 + 3.
// Lane l owns tile l&31, channel half l>>5 (A operand) / output channel l&31 (B operand, result).
//   * No two waves of a group use the same filter taps: the transformed filter never touches LDS, each lane loads its taps as
//     8-byte pieces straight from the L2-resident panel (vv_pack_wino44: [36][Kp/8][2 sub-steps][2 halves][N][2]) into the registers
//     the two MFMAs of that tap have just released, one sub-step ahead.
//   * An 8-channel chunk is multiplied in two sub-steps of 4 channels (lane half h: channels 4h+2s, 4h+2s+1), so that the input
//     transform and the taps of a sub-step take 12 + 12 registers instead of 24 + 24.  The wave's row of B^T d B is a compile-time
//     constant of the K loop (one copy of the loop per wave role): immediate LDS offsets, literal coefficients.
//   * The halo tile [NI][HH][HW] goes global -> registers -> LDS with the producer's BatchNorm + ReLU applied on the way in (as in
//     vv_wino.hip), requested in the middle of the chunk before; LDS image: four planes (half, sub-step) of 8-byte slots, columns
//     split by x mod 4 and row / image strides padded per level so that the 16 lanes a ds_read2_b64 access services together hit 16
//     different bank pairs (checked at compile time).
//   * Epilogue: every wave applies the column half of A^T . A to its xi; the six waves meet in LDS (two rounds of eight accumulator
//     registers) and finish (register, output-row pair) units: bias, ReLU (eval), NHWC buffer stores, BatchNorm sum / sum of squares
//     or the fused first pass of the consumer's BatchNorm backward -- the same contract as wino_conv_kernel.
// Where it stands (round 5): results agree with the direct kernel / float64 to 1 - 3e-6 of the tensor maximum on every
// level (tests/test_gpu_wino44.py; F(2x2): 4e-7 - 1.7e-6); matrix pipe busy 0.33 - 0.45 (F(2x2): 0.6 - 0.7) -- per MFMA it issues ~8 other
// instructions (F(2x2): ~4) and an fp32 MFMA is only 2 x the packed vector rate -- so it is faster than F(2x2) where GEMM-K >= 64 and its
// coarse workgroups fill the chip in whole rounds (-7 ... -26 % per launch).  UNetBank routes the data-gradient launches of a train step
// (their rounding feeds no gate and no statistic: gradient statistics unchanged to three digits) and the eval-mode forward's launches
// to it by that policy; forward launches of a train step only on request (VV_WINO44=1 | all).
#include <type_traits>
#include "vv_common.h"
namespace {

constexpr int W4N = 384;               // threads per sub-group: one wave per xi
constexpr int W4G = 2;                 // sub-groups (pixel tiles) per workgroup -- see the kernel's header
constexpr int W4T = 32;                // tiles per workgroup
constexpr int SB4_MASK = 0x386;        // may cross a scheduling barrier: VALU, SALU, LDS -- not MFMA, not VMEM
// Interpolation points of the Toom-Cook construction: (0, +-PA, +-PB, infinity).  The textbook choice (0, +-1, +-2) scaled by 3/4:
// the same zero pattern and +- symmetry of the three matrices, every constant a dyadic rational (exact in fp32), and a 2.4 x smaller
// rounding error of the fp32 transforms (single tile, 64 channels: mean 8.3e-7 of the maximum against 2.0e-6; 19 symmetric and
// asymmetric point sets were compared on the CPU, profiles/r05_wino44_points.txt):
//   B^T = [a2b2 0 -(a2+b2) 0 1 0; 0 -a b2 -b2 a 1 0; 0 a b2 -b2 -a 1 0; 0 -a2 b -a2 b 1 0; 0 a2 b -a2 -b 1 0; 0 a2b2 0 -(a2+b2) 0 1]
//   A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a2 a2 b2 b2 0; 0 a3 -a3 b3 -b3 1]
//   G   = [1/N0 0 0; (1 +-a a2)/Na; (1 +-b b2)/Nb; 0 0 1],  N0 = a2 b2, Na = 2 a2 (a2 - b2), Nb = 2 b2 (b2 - a2)
constexpr float PA = 0.75f, PB = 1.5f;
constexpr float PA2 = PA * PA, PB2 = PB * PB, PA3 = PA2 * PA, PB3 = PB2 * PB;
constexpr float PA2B2 = PA2 * PB2, PS2 = PA2 + PB2, PAB2 = PA * PB2, PA2B = PA2 * PB;

// LDS hand-over between the waves of a workgroup WITHOUT the vector-memory drain a __syncthreads() can carry (halo and tap loads stay
// in flight across it); the empty asm keeps the compiler from lifting later LDS reads above the barrier
__device__ __forceinline__ void vv_lds_barrier4() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int H_>
struct W4Geo {
  static constexpr int TPI = H_ / 4;                       // tiles per image side
  static constexpr int TP = TPI * TPI;                     // tiles per image
  static constexpr int TPW = TP < W4T ? TP : W4T;          // tiles of one image handled by one workgroup
  static constexpr int NI = W4T / TPW;                     // images per workgroup
  static constexpr int PARTS = TP / TPW;                   // workgroups per image
  static constexpr int TROWS = TPW / TPI;                  // tile rows per part
  static constexpr int HH = 4 * TROWS + 2, HW = H_ + 2;
  static constexpr int HWQ = (HW + 3) / 4;                 // slots of one (x mod 4) column class per halo row
  // 8-byte slots per halo row / per image, padded against LDS bank conflicts of the patch reads.  The compiler pairs them into
  // ds_read2_b64, which is serviced in groups of 16 consecutive lanes over 32 banks ((a / 4) mod 32): the 16 tiles of such a group
  // must lie in 16 different slots mod 16 (ok() below).  (The first layout was padded for ds_read_b64's 32-lane / 64-bank rule: on the
  // 8x8 level SQ_LDS_BANK_CONFLICT was 0.47 of the LDS-active cycles.)
  static constexpr int ROW = H_ == 32 ? 38 : (H_ == 16 ? 21 : (H_ == 8 ? 14 : 8));
  static constexpr int IMG = H_ == 32 ? HH * ROW : (H_ == 16 ? 400 : (H_ == 8 ? 146 : 49));
  static constexpr int PLANE = NI * IMG;
  static constexpr int tile_slot(const int l) {            // lane-dependent part of a patch read's slot
    const int tim = l / TPW, rem = l % TPW;
    return tim * IMG + 4 * (rem / TPI) * ROW + rem % TPI;
  }
  static constexpr bool ok() {
    if (ROW < 4 * HWQ || IMG < HH * ROW) return false;
    for (int l0 = 0; l0 < 32; l0 += 16) {
      unsigned seen = 0;
      for (int l = l0; l < l0 + 16; ++l) {
        const unsigned bit = 1u << (tile_slot(l) & 15);
        if (seen & bit) return false;
        seen |= bit;
      }
    }
    return true;
  }
};
static_assert(W4Geo<32>::ok() && W4Geo<16>::ok() && W4Geo<8>::ok() && W4Geo<4>::ok(), "LDS image: bank conflicts or overlap");

// NS ("N share"): the two groups of a workgroup take the two N tiles of ONE pixel tile and share its halo image -- every thread of the
// workgroup stages half as many items, the halo crosses L2 -> LDS once for 64 output channels; NS = false: two pixel tiles of one N tile
// (launches with 32 output channels).
template <int H_, bool NS>
__global__ void __launch_bounds__(W4N * W4G, 3)
wino44_conv_kernel(const vv_conv_params p, const int NT, const int NN, const int total, const int nper) {
  using G_ = W4Geo<H_>;
  constexpr int TPI = G_::TPI, TPW = G_::TPW, NI = G_::NI, PARTS = G_::PARTS, HH = G_::HH, HW = G_::HW, HWQ = G_::HWQ;
  constexpr int ROW = G_::ROW, IMG = G_::IMG, PLANE = G_::PLANE;
  constexpr int CK = 8, Q = 2;
  constexpr int HY0 = PARTS == 1 ? 1 : 0, HR = PARTS == 1 ? HH - 2 : HH;      // halo rows that can lie inside an image
  constexpr int NITEMS = NI * HR * H_ * Q;
  constexpr int STN = NS ? W4N * W4G : W4N;                // threads that stage one halo image
  constexpr int NIT = (NITEMS + STN - 1) / STN;
  constexpr int HALO8 = 4 * PLANE;                         // 8-byte slots of the halo image
  constexpr int EX8 = 6 * 8 * 64 * 2;                      // epilogue exchange: [wave][8 regs][64 lanes] float4
  constexpr int L8 = HALO8 > EX8 ? HALO8 : EX8;
  // A workgroup is TWO such six-wave groups on neighbouring pixel tiles of one (UNet, N tile), each with its own LDS image; they
  // share nothing but the barriers.  Why: as workgroups of their own only one fitted a CU (measured: half the expected occupancy,
  // matrix pipe busy 0.31) -- six waves land 2 + 2 + 1 + 1 on the four SIMDs and a second set of six does not fit the register file
  // beside them; twelve waves of one workgroup land 3 + 3 + 3 + 3.
  __shared__ v2f lds_all[W4G][L8 + 6 * 32 + 256];          // + [2][6 waves][32] BatchNorm partials + [2][<= 256] scale / shift of the input's BatchNorm
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int sg = wv / 6;
  v2f* const lds8 = lds_all[sg];
  float* lds = reinterpret_cast<float*>(lds8);
  v2f* const halo = NS ? lds_all[0] : lds8;

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  int nn, pt, g;
  bool live = true;
  if constexpr (NS) {
    const int NNW = NN / W4G;            // N-tile pairs; NN is even (host)
    nn = (w % NNW) * W4G + sg; w /= NNW;
    pt = w % NT;
    g = w / NT;
  } else {
    nn = w % NN; w /= NN;                // the N tiles of one pixel tile are neighbours in launch order: they share the halo in L2
    const int NTP = (NT + W4G - 1) / W4G;
    const int pt_ = (w % NTP) * W4G + sg;
    live = pt_ < NT;                     // (odd tile count: the last workgroup's second group repeats the last tile and writes nothing)
    pt = live ? pt_ : NT - 1;
    g = w / NTP;
  }

  const int tid = threadIdx.x - sg * W4N, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int stid = NS ? (int)threadIdx.x : tid;           // this thread among those that stage the image
  const int xi = wv - sg * 6;
  const int img0 = (pt / PARTS) * NI, part = pt % PARTS;
  const int y0 = part * (4 * G_::TROWS) - 1;               // conv-input row of halo row 0 (column origin is -1)
  const VVSrc s = vv_make_src(p, g, H_, H_);
  const int Cout = p.Cout, CinP = p.CinP, KQ = CinP >> 3;
  const int co0 = nn * 32;
  const float* __restrict__ wg = p.w + (int64_t)g * p.w_gstride;

  // ---- halo staging.  Items = (pixel, channel quad) of the pixels that can lie inside an image: the halo columns 0 and HW - 1
  //      (every workgroup spans the image width) and, where a workgroup spans the image height, the halo rows 0 and HH - 1 are zero
  //      padding for every chunk -- zeroed once, never written again (three items per thread on every level).  An item travels
  //      global -> registers -> LDS like in vv_wino.hip (raw buffer loads, out-of-image items out of range, the producer's BatchNorm +
  //      ReLU on the way into LDS), with fewer registers held across the MFMA phase: per item ONE packed word = image slot << 16 |
  //      pixel offset inside the tile, buffer offsets rebuilt per chunk; the producer's scale / shift of every input channel wait in
  //      LDS instead of in eight registers.  (An LDS-DMA landing zone was built and dropped: with it the tap loads had to become
  //      inline asm with hand-counted waits, and the compiler copies / spills a register it believes valid while the load is in flight.)
  unsigned valid = 0;
  unsigned itm[NIT];
  const int tile = (img0 * H_ + y0) * H_ - 1;
  const int q4 = (stid % Q) * 4;
  for (int hp = stid; hp < NI * HH * HW; hp += STN) {
    const int hx = hp % HW, t = hp / HW;
    const int hy = t % HH, im = t / HH;
    if (hx == 0 || hx == HW - 1 || hy < HY0 || hy >= HY0 + HR) {
      const int sl = im * IMG + hy * ROW + (hx & 3) * HWQ + (hx >> 2);
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) halo[pl * PLANE + sl] = (v2f){0.f, 0.f};
    }
  }
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = stid + k * STN;
    const int q = it % Q, hp = it / Q;
    const int hx = hp % H_ + 1, t = hp / H_;
    const int hy = t % HR + HY0, im = t / HR;
    const bool inr = NITEMS % STN == 0 || it < NITEMS;
    const int y = y0 + hy;
    const bool ok = inr && (unsigned)y < (unsigned)H_ && (img0 + im) < s.B;
    valid |= ok ? (1u << k) : 0u;
    const int pixv = (im * H_ + hy) * H_ + hx;                                                  // < 2^16 on every level
    const int sl = inr ? q * 2 * PLANE + im * IMG + hy * ROW + (hx & 3) * HWQ + (hx >> 2) : 0xFFFF;  // plane (q, 0); (q, 1) is PLANE further
    itm[k] = ((unsigned)sl << 16) | (unsigned)pixv;
  }
  static_assert((NI * H_ + HH) * H_ + HW < 65536 && 4 * PLANE < 65535, "packed item word");
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned bvo = (unsigned)(half * Cout + co0 + l31) * 8u;                 // this lane inside a [2 halves][Cout] float2 slab
  const int bsub = 2 * Cout * 8;                                                 // bytes between the sub-step slabs of a chunk
  const int bnu = KQ * 2 * bsub;                                                 // bytes between nu panels
  const int bxi = xi * 6 * bnu;
  const int nact = s.mode == VV_IN_ACT ? CinP : (s.mode == VV_IN_CAT ? s.csplit : 0);   // channels [0, nact) carry a BatchNorm + ReLU
  float4* const ab4 = reinterpret_cast<float4*>((NS ? lds_all[0] : lds8) + L8 + 6 * 32);      // [2][CinP / 4]: scale, shift
  for (int i = stid; i < (nact >> 2); i += STN) {
    ab4[i] = *reinterpret_cast<const float4*>(s.a + 4 * i);
    ab4[(CinP >> 2) + i] = *reinterpret_cast<const float4*>(s.b + 4 * i);
  }
  // Order of a wave's vector-memory instructions (they return in order, one counter): the taps of sub-step 1 go out during sub-step
  // 0, THEN the halo loads of the next chunk, then -- during sub-step 1 -- the taps of the next chunk's sub-step 0: the compiler's
  // counted waits in front of sub-step 1's MFMAs leave the halo loads in flight, the wait in front of commit() leaves the taps.
  float4 r[NIT];
  auto issue = [&](const int c0) {
    const bool second = __builtin_amdgcn_readfirstlane((int)((s.mode == VV_IN_CAT) && c0 >= s.csplit)) != 0;
    const float* base = second ? s.p1 + s.co1 : s.p0 + s.co0;
    const int cs = second ? s.cs1 : s.cs0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
    const int soff = (second ? c0 - s.csplit : c0) * 4;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      // (branch-free: bit 31 set = out of range = zeros; a ?: here became a branch around every load)
      const unsigned vo_ = ((unsigned)((tile + (int)(itm[k] & 0xFFFFu)) * cs + q4) * 4u) |
                           ((((valid >> k) & 1u) ^ 1u) << 31);
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo_, soff, 0);
      r[k] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  auto commit = [&](const int c0) {
    const bool act = c0 < nact;                              // (chunks never straddle the concat split)
    float4 sa = make_float4(1.f, 1.f, 1.f, 1.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
      sa = ab4[(c0 + q4) >> 2];
      sb = ab4[(CinP >> 2) + ((c0 + q4) >> 2)];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int sl = (int)(itm[k] >> 16);
      if (NITEMS % STN == 0 || k < NIT - 1 || sl != 0xFFFF) {
        float4 v = r[k];
        if (act) {                                             // (uniform; the per-item part is a select: invalid items stay zero)
          const float4 a = vv_act4(v, sa, sb);
          const bool vk = (valid >> k) & 1u;
          v = make_float4(vk ? a.x : v.x, vk ? a.y : v.y, vk ? a.z : v.z, vk ? a.w : v.w);
        }
        halo[sl] = (v2f){v.x, v.y};
        halo[sl + PLANE] = (v2f){v.z, v.w};
      }
    }
  };
  v2f u[6];
  auto load_u = [&](const int step, const int n) -> v2f {    // step = chunk * 2 + sub-step
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsW, bvo, bxi + n * bnu + step * bsub, 0));
  };

  // ---- this lane's tile and its patch origin in LDS
  const int pbase = half * 2 * PLANE + G_::tile_slot(l31);

  v16f acc[6];
  const int nsteps = 2 * KQ;
  // one sub-step: input transform of this wave's row (18 / 24 LDS reads, <= 36 packed VALU), 12 MFMAs, the taps of the next sub-step
  // loaded into the registers the second k step has just released.  XI is a compile-time constant (the K loop below exists once per
  // wave role): the rows of B^T become immediate LDS offsets and literal constants --
  //   (rows of B^T above; with the textbook points: 0: 4 d0 - 5 d2 + d4, 1: -4 d1 - 4 d2 + d3 + d4, ..., 5: 4 d1 - 5 d3 + d5)
  // LAST: the sub-step that ends the K loop loads nothing (a load still in flight into a register the compiler considers dead would
  // land in whatever the epilogue keeps there).
  auto substep = [&](const auto XI_, const int step, const auto SUB_, const auto first, const auto LAST_) {
    constexpr int XI = decltype(XI_)::value, sub = decltype(SUB_)::value;
    constexpr bool LAST = decltype(LAST_)::value;
    const int nxt = step + 1;
    // (the patch origin is made opaque per sub-step: left to itself the compiler hoists eight or nine derived LDS addresses out of the
    //  K loop -- ds_read2_b64 reaches 2 KB -- and spills the item words to make room)
    int pb = pbase;
    asm volatile("" : "+v"(pb));
    const v2f* const pp = halo + pb + sub * PLANE;
    v2f R[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int o = (b & 3) * HWQ + (b >> 2);
      if constexpr (XI == 0) {
        R[b] = PA2B2 * pp[o] - PS2 * pp[2 * ROW + o] + pp[4 * ROW + o];
      } else if constexpr (XI == 5) {
        R[b] = PA2B2 * pp[ROW + o] - PS2 * pp[3 * ROW + o] + pp[5 * ROW + o];
      } else {
        const v2f d1 = pp[ROW + o], d2 = pp[2 * ROW + o], d3 = pp[3 * ROW + o], d4 = pp[4 * ROW + o];
        if constexpr (XI == 1) R[b] = (d4 - PB2 * d2) + (PA * d3 - PAB2 * d1);
        if constexpr (XI == 2) R[b] = (d4 - PB2 * d2) - (PA * d3 - PAB2 * d1);
        if constexpr (XI == 3) R[b] = (d4 - PA2 * d2) + (PB * d3 - PA2B * d1);
        if constexpr (XI == 4) R[b] = (d4 - PA2 * d2) - (PB * d3 - PA2B * d1);
      }
    }
    v2f V[6];
    {
      const v2f pq = R[4] - PB2 * R[2], qq = PA * R[3] - PAB2 * R[1], rr = R[4] - PA2 * R[2], ss = PB * R[3] - PA2B * R[1];
      V[0] = PA2B2 * R[0] - PS2 * R[2] + R[4];
      V[1] = pq + qq;
      V[2] = pq - qq;
      V[3] = rr + ss;
      V[4] = rr - ss;
      V[5] = PA2B2 * R[1] - PS2 * R[3] + R[5];
    }
    __builtin_amdgcn_sched_barrier(SB4_MASK);
    // tap n has landed (the asm ties the wait to the register the MFMA reads); written out six times: the count must be a literal
#define W4_X(n)                                                                                        \
    if constexpr (decltype(first)::value) {                                                             \
      const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   \
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].x, u[n].x, z, 0, 0, 0);                         \
    } else {                                                                                            \
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].x, u[n].x, acc[n], 0, 0, 0);                    \
    }
#define W4_Y(n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].y, u[n].y, acc[n], 0, 0, 0);
    // nu-major: both k steps of a tap back to back, then the tap's reload -- every tap is requested ten MFMAs + one input transform
    // ahead of its next use (k-step-major order left the last tap four MFMAs: every sub-step waited for an L2 round trip)
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      W4_X(n)
      W4_Y(n)
      __builtin_amdgcn_sched_barrier(SB4_MASK);
      if constexpr (!LAST) u[n] = load_u(nxt, n);
      __builtin_amdgcn_sched_barrier(SB4_MASK);
    }
#undef W4_X
#undef W4_Y
  };
  const std::true_type yes{};
  const std::false_type no{};
  const std::integral_constant<int, 0> sub0{};
  const std::integral_constant<int, 1> sub1{};
  // the K loop of one wave role (every role passes the same barriers).  Order of a wave's vector-memory instructions (they return in
  // order, one counter): the taps of sub-step 1 go out during sub-step 0, THEN the halo loads of the next chunk, then -- during
  // sub-step 1 -- the taps of the next chunk's sub-step 0: the compiler's counted waits in front of sub-step 1's MFMAs leave the
  // halo loads in flight, the wait in front of commit() (six taps younger) leaves the taps in flight.
  auto kloop = [&](const auto XI_) {
    if (KQ == 1) {
      substep(XI_, 0, sub0, yes, no);
      substep(XI_, 1, sub1, no, yes);
      return;
    }
    auto next_chunk = [&](const int kq) {
      vv_lds_barrier4();                  // every wave finished reading the previous chunk
      commit(kq * CK);
      vv_lds_barrier4();
    };
    substep(XI_, 0, sub0, yes, no);
    issue(CK);
    substep(XI_, 1, sub1, no, no);
    // (the last chunk is peeled: one loop body, no branch inside it -- with one the accumulators were copied and spilled at the join)
    for (int kq = 1; kq < KQ - 1; ++kq) {
      next_chunk(kq);
      substep(XI_, 2 * kq, sub0, no, no);
      issue((kq + 1) * CK);
      substep(XI_, 2 * kq + 1, sub1, no, no);
    }
    next_chunk(KQ - 1);
    substep(XI_, 2 * KQ - 2, sub0, no, no);
    substep(XI_, 2 * KQ - 1, sub1, no, yes);
  };

  issue(0);
#pragma unroll
  for (int n = 0; n < 6; ++n) u[n] = load_u(0, n);
  __syncthreads();                      // the scale / shift table and the zero borders are complete
  commit(0);
  __syncthreads();
  switch (xi) {
    case 0: kloop(std::integral_constant<int, 0>{}); break;
    case 1: kloop(std::integral_constant<int, 1>{}); break;
    case 2: kloop(std::integral_constant<int, 2>{}); break;
    case 3: kloop(std::integral_constant<int, 3>{}); break;
    case 4: kloop(std::integral_constant<int, 4>{}); break;
    default: kloop(std::integral_constant<int, 5>{}); break;
  }

  // ---- epilogue.  Columns in registers (M = this wave's xi, indexed by nu):
  //        T0 = M0 + M1 + M2 + M3 + M4    T1 = a (M1 - M2) + b (M3 - M4)    T2 = a2 (M1 + M2) + b2 (M3 + M4)    T3 = a3 (M1 - M2) + b3 (M3 - M4) + M5
  //      rows across the six waves through LDS, the same combination of T(xi).  Accumulator register i of lane (half, l31) belongs to
  //      tile 8 (i / 4) + 4 half + i % 4: the wave-uniform part of a unit's address uses the tile of half 0, the lanes of half 1 carry
  //      the distance of four tiles (LP pixels) in their offsets.
  constexpr int LP = TPI == 8 ? 16 : (TPI == 4 ? 4 * H_ : (TPI == 2 ? H_ * H_ : 4 * H_ * H_));
  constexpr int HIMG = TPW == 4 ? 1 : (TPW == 1 ? 4 : 0);                // images between the two lane halves
  const bool bnf = p.bn_partial != nullptr;
  float bna = 0.f, bnb = 0.f, bni = 0.f, bnm = 0.f;
  __amdgpu_buffer_rsrc_t rsZ = rsW;
  int vz = 0;
  if (bnf) {
    const int64_t bo = (int64_t)g * p.bn_gstride + co0 + l31;
    bna = p.bn_a[bo]; bnb = p.bn_b[bo]; bni = p.bn_invstd[bo];
    bnm = -p.bn_mean[bo] * bni;                                            // xhat = z invstd - mean invstd
    rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bn_z + (int64_t)g * p.bn_z_gstride), 0, 0x7FFFFFFF, 0x00020000);
    vz = (half * LP * Cout + co0 + l31) * 4;
  }
  const float bias = p.bias ? p.bias[(int64_t)g * p.bias_gstride + co0 + l31] : 0.f;
  const bool relu = (p.pad0 & VV_CONV_RELU) != 0;                          // BatchNorm folded into the filter (eval)
  const int ocs = p.out.cstride;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
      p.out.ptr + (int64_t)g * p.out.gstride + p.out.coff, 0, 0x7FFFFFFF, 0x00020000);
  const int vo = (half * LP * ocs + co0 + l31) * 4;
  v4f* ex4 = reinterpret_cast<v4f*>(lds8);
  v2f s12 = {0.f, 0.f}, q12 = {0.f, 0.f};
  __syncthreads();                      // all MFMA-phase LDS reads done: LDS becomes the exchange buffer
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    if (rd) __syncthreads();            // round 0's exchange fully read
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
      const int i = rd * 8 + ii;
      const float m0 = acc[0][i], m1 = acc[1][i], m2 = acc[2][i], m3 = acc[3][i], m4 = acc[4][i], m5 = acc[5][i];
      const float s12_ = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      const v4f t = {m0 + s12_ + s34, PA * d12 + PB * d34, PA2 * s12_ + PB2 * s34, PA3 * d12 + PB3 * d34 + m5};
      ex4[(xi * 8 + ii) * 64 + lane] = t;
    }
    // this wave's units of the round: u = (register ii, output-row pair rp); three or two per wave, rotated between the rounds
    const int ustart = rd == 0 ? xi : (xi >= 2 ? xi - 2 : xi + 4);
    float zq[3][8];
    bool jok[3];
    int so_[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int un = ustart + 6 * k;                                 // wave-uniform
      const int ii = un >> 1, rp = un & 1;
      const int i = rd * 8 + ii;
      const int t2 = 8 * (i >> 2) + (i & 3);
      const int im = t2 / TPW, rem = t2 % TPW;
      const int oy = 4 * (part * G_::TROWS + rem / TPI) + 2 * rp, ox = 4 * (rem % TPI);
      jok[k] = live && un < 16 && (img0 + im + HIMG * half) < p.B;
      so_[k] = ((img0 + im) * H_ + oy) * H_ + ox;                    // pixel index of the unit's first output (half 0)
      if (bnf) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          zq[k][e] = jok[k] ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                  rsZ, vz, (so_[k] + (e >> 2) * H_ + (e & 3)) * Cout * 4, 0))
                            : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int un = ustart + 6 * k;
      if (un < 16) {                                                 // wave-uniform
        const int ii = un >> 1, rp = un & 1;
        const v4f* e = ex4 + ii * 64 + lane;
        const v4f t1 = e[1 * 512], t2v = e[2 * 512], t3 = e[3 * 512], t4 = e[4 * 512];
        const v4f sA = t1 + t2v, dA = t1 - t2v, sB = t3 + t4, dB = t3 - t4;
        v4f ya, yb;
        if (rp == 0) {
          ya = e[0] + sA + sB;
          yb = PA * dA + PB * dB;
        } else {
          ya = PA2 * sA + PB2 * sB;
          yb = PA3 * dA + PB3 * dB + e[5 * 512];
        }
        ya += bias;
        yb += bias;
        if (relu) {     // compare + select, not v_max: a diverged model's NaN stays a NaN, like torch's ReLU
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ya[j] = ya[j] < 0.f ? 0.f : ya[j];
            yb[j] = yb[j] < 0.f ? 0.f : yb[j];
          }
        }
        if (jok[k]) {
          const int so = so_[k] * ocs * 4;
          {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ya[j]), rsO, vo, so + j * ocs * 4, 0);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yb[j]), rsO, vo, so + (H_ + j) * ocs * 4, 0);
            }
          }
          if (bnf) {
            // dz = dA [a z + b > 0];  partial sums of dz and dz * xhat  (bn_bwd_reduce_kernel<.., 0>, fused)
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              const v2f za = {zq[k][j], zq[k][j + 1]}, zb = {zq[k][4 + j], zq[k][4 + j + 1]};
              const v2f pa = bna * za + bnb, pb = bna * zb + bnb;
              const v2f da = {pa.x > 0.f ? ya[j] : 0.f, pa.y > 0.f ? ya[j + 1] : 0.f};
              const v2f db = {pb.x > 0.f ? yb[j] : 0.f, pb.y > 0.f ? yb[j + 1] : 0.f};
              s12 += da + db;
              q12 = __builtin_elementwise_fma(da, bni * za + bnm, q12);
              q12 = __builtin_elementwise_fma(db, bni * zb + bnm, q12);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              const v2f a2 = {ya[j], ya[j + 1]}, b2 = {yb[j], yb[j + 1]};
              s12 += a2 + b2;
              q12 = __builtin_elementwise_fma(a2, a2, q12);
              q12 = __builtin_elementwise_fma(b2, b2, q12);
            }
          }
        }
      }
    }
  }
  float* const sout = bnf ? p.bn_partial : p.stats;
  if (sout) {
    float s1 = s12.x + s12.y, s2 = q12.x + q12.y;
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    float* sp = lds + L8 * 2;
    if (half == 0) {
      sp[xi * 32 + l31] = s1;
      sp[(6 + xi) * 32 + l31] = s2;
    }
    __syncthreads();
    if (tid < 32 && live) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        t1 += sp[k * 32 + tid];
        t2 += sp[(6 + k) * 32 + tid];
      }
      float* st = sout + ((int64_t)(g * NT + pt) * 2) * Cout + co0 + tid;
      st[0] = t1;
      st[Cout] = t2;
    }
  }
}

// U = G g G^T of every (ci, co) filter, in the B-operand panel layout [xi*6+nu][Kp/8][2 sub-steps][2 halves][N][2]:
// element (t, k, n) at  ((((t * Kp/8 + k/8) * 2 + (k%4)/2) * 2 + (k%8)/4) * N + n) * 2 + k%2.
//   (G as in the table at the top of this file)
// mode 0: forward       g[a][b] = W[co = n][ci = k][a][b]
// mode 1: data gradient g[a][b] = W[co = k][ci = n][2-a][2-b]
__global__ void __launch_bounds__(VV_WG)
wino44_pack_kernel(const vv_pack_entry* __restrict__ table, const float* __restrict__ params, const int64_t params_gstride,
                   float* __restrict__ packed, const int64_t packed_gstride) {
  // one workgroup = one 8-channel K group (kq) x 32 output channels: 256 (k, n) filters, one per thread; the 36 transformed taps are
  // exchanged through LDS so that every tap leaves as four contiguous 256-byte runs [sub][half][n][2]
  __shared__ float ex[36][256];
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
    // rows of G as (x, y, z) -> G[i][0] x + G[i][1] y + G[i][2] z
    // (in double: the filter transform then contributes one final rounding, whatever the constants)
    auto grow = [](const int i, const double x, const double y, const double z) -> double {
      constexpr double a = PA, b = PB, a2 = a * a, b2 = b * b;
      constexpr double n0 = a2 * b2, na = 2.0 * a2 * (a2 - b2), nb = 2.0 * b2 * (b2 - a2);
      switch (i) {
        case 0: return x / n0;
        case 1: return (x + a * y + a2 * z) / na;
        case 2: return (x - a * y + a2 * z) / na;
        case 3: return (x + b * y + b2 * z) / nb;
        case 4: return (x - b * y + b2 * z) / nb;
        default: return z;
      }
    };
    double rr[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int b = 0; b < 3; ++b) rr[i][b] = grow(i, gk[0][b], gk[1][b], gk[2][b]);
    const int hf = kl >> 2, sub = (kl & 3) >> 1, j = kl & 1;
    const int slot = (sub * 2 + hf) * 64 + nl * 2 + j;            // [sub][half][n][j]
    __syncthreads();                                               // previous block's exchange fully read
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
      for (int nu = 0; nu < 6; ++nu) ex[xi * 6 + nu][slot] = (float)grow(nu, rr[xi][0], rr[xi][1], rr[xi][2]);
    __syncthreads();
    // thread t = position inside the tap's [sub][half][32 n][2] block of this (kq, nb): runs of 64 floats per (sub, half)
    const int sh = t >> 6, rem = t & 63;
#pragma unroll
    for (int xn = 0; xn < 36; ++xn)
      dst[((((int64_t)xn * KQ + kq) * 4 + sh) * e.N + nb * 32) * 2 + rem] = ex[xn][t];
  }
}

template <int H_>
int launch_wino44(const vv_conv_params* p, hipStream_t st) {
  using G_ = W4Geo<H_>;
  const int NT = ((p->B + G_::NI - 1) / G_::NI) * G_::PARTS;
  const int NN = p->Cout / 32;
  if (NN % W4G == 0) {                 // the groups of a workgroup = the two N tiles of one pixel tile (shared halo image)
    const int total = p->G * (NN / W4G) * NT;
    const int nper = (total + 7) / 8;
    VV_LAUNCH((wino44_conv_kernel<H_, true>), dim3(nper * 8), dim3(W4N * W4G), 0, st, *p, NT, NN, total, nper);
  } else {                             // ... = two pixel tiles of one N tile
    const int total = p->G * NN * ((NT + W4G - 1) / W4G);
    const int nper = (total + 7) / 8;
    VV_LAUNCH((wino44_conv_kernel<H_, false>), dim3(nper * 8), dim3(W4N * W4G), 0, st, *p, NT, NN, total, nper);
  }
  VV_CHECK_LAUNCH();
  return VV_OK;
}

}  // namespace

extern "C" int vv_wino44_ntiles(int32_t B, int32_t H) {
  switch (H) {
    case 32: return ((B + W4Geo<32>::NI - 1) / W4Geo<32>::NI) * W4Geo<32>::PARTS;
    case 16: return ((B + W4Geo<16>::NI - 1) / W4Geo<16>::NI) * W4Geo<16>::PARTS;
    case 8: return ((B + W4Geo<8>::NI - 1) / W4Geo<8>::NI) * W4Geo<8>::PARTS;
    case 4: return ((B + W4Geo<4>::NI - 1) / W4Geo<4>::NI) * W4Geo<4>::PARTS;
  }
  return -1;
}

extern "C" int vv_conv_wino44(const vv_conv_params* p, vv_stream stream) {
  if (!p || !p->src0.ptr || !p->w || !p->out.ptr) return VV_ERR_BAD_ARG;
  if (p->G <= 0 || p->B <= 0 || p->kind != VV_CONV3 || p->H != p->W) return VV_ERR_BAD_ARG;
  if (p->Cout % 32 || p->CinP % 8 || p->CinP > 256) return VV_ERR_UNSUPPORTED;      // (256: the scale / shift table in LDS)
  if (p->out1.ptr) return VV_ERR_UNSUPPORTED;                          // second output view: bf16-output launches of vv_conv_mfma
  if (p->bn_partial && (p->stats || !p->bn_z || !p->bn_a || !p->bn_b || !p->bn_mean || !p->bn_invstd)) return VV_ERR_BAD_ARG;
  {
    // the epilogue addresses one UNet's output (and z) with 32-bit byte offsets
    const int64_t px = (int64_t)p->B * p->H * p->W * 4;
    const int64_t cs = p->out.cstride > p->Cout ? p->out.cstride : p->Cout;
    if (px * cs >= (1ll << 31)) return VV_ERR_UNSUPPORTED;
  }
  if (p->in_mode != VV_IN_PLAIN && p->in_mode != VV_IN_ACT && p->in_mode != VV_IN_CAT)
    return VV_ERR_UNSUPPORTED;    // feed the materialised tensor (vv_pool_act / vv_cube_erase) as VV_IN_PLAIN
  if (p->in_mode == VV_IN_CAT && p->csplit % 8) return VV_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  switch (p->H) {
    case 32: return launch_wino44<32>(p, st);
    case 16: return launch_wino44<16>(p, st);
    case 8: return launch_wino44<8>(p, st);
    case 4: return launch_wino44<4>(p, st);
  }
  return VV_ERR_UNSUPPORTED;
}

extern "C" int vv_pack_wino44(const vv_pack_entry* table_dev, int32_t nentries, int32_t G, const float* params,
                              int64_t params_gstride, float* packed, int64_t packed_gstride, int32_t max_kn,
                              vv_stream stream) {
  if (!table_dev || !params || !packed || nentries <= 0) return VV_ERR_BAD_ARG;
  int bx = (max_kn + VV_WG - 1) / VV_WG;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  VV_LAUNCH(wino44_pack_kernel, dim3(bx, nentries, G), dim3(VV_WG), 0, (hipStream_t)stream, table_dev, params, params_gstride,
            packed, packed_gstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
