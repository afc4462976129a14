// Weight gradient of the 3x3 / stride 1 / pad 1 convolution with bf16 operands (mixed precision, BASELINE config 4):
//     dW[tap][ci][co] = sum_pixels bf16(act[pixel + tap][ci]) * bf16(dy[pixel][co]),   fp32 accumulation
// on v_mfma_f32_32x32x16_bf16: M = ci, N = co, K = 16 pixels per instruction.  The instruction wants, per lane, 8 consecutive
// K values (= 8 pixels) of ONE channel, while the tensors are NHWC (channels contiguous).  The staged tiles stay pixel-major
// ([pixel][32 channels] bf16, 64 B per pixel, written by the register-staged loader with the producer's BatchNorm+ReLU applied on
// the way in) and the operands come out of LDS already transposed: gfx950's ds_read_b64_tr_b16 lets the 16 lanes of a group
// address a 4-pixel x 16-channel block (lane 4j+q: pixel j, channels 4q..4q+3) and hands lane l the 4 pixels of channel l, so an
// operand is two such reads with no VALU work (round 2 gathered dwords and packed pixel pairs with 16 v_perm_b32 per 9 MFMAs:
// the matrix pipe was busy 0.18-0.27 of the time).  Per-lane addresses are free, so tap shifts are immediate offsets on one base
// register; the three row shifts are still a sliding window over the rows a wave walks down: 6 + 2 LDS reads per 9 MFMAs.
//
// The matrix work is 1/16 of the fp32 instruction's, so this kernel is bound by memory and its first concern is not to re-read:
// one workgroup stages up to 64 input channels x 64 output channels per pixel tile (each tensor crosses HBM once when the
// layer has <= 64 channels) and its four waves own the (ci-block, co-block) pairs -- or split the pixels when the layer has
// fewer blocks, and are then summed through LDS in wave order -- so every wave keeps its own 9 x (32 x 32) accumulators.
// Slabs leave through vv_wgrad_reduce like those of the fp32 kernels (fixed order: bitwise reproducible).
// DY16 / A16 (compile-time): dy / the layer input are bf16 tensors in HBM (what the mixed-precision bank stores) -- 8-byte items,
// copied (dy, plain inputs) or unpacked -> BatchNorm+ReLU -> repacked (pre-BN inputs) on their way into LDS.
//
// Replaces the autograd weight gradient of nn.Conv2d(k3, p1) (model/unet.py:10,13; cuDNN in the reference) under
// torch.autocast(bfloat16)-style operand rounding.
#include "vv_common.h"

namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* vv_lds_t;          // byte pointer into LDS

__device__ __forceinline__ vv_lds_t vv_lds_ptr(float* p) { return (vv_lds_t)p; }

// One MFMA operand (8 consecutive K = pixels of this lane's channel) from a pixel-major bf16 tile: two ds_read_b64_tr_b16.
// Within a 16-lane group, lane 4j+q supplies the address of [pixel j][4 channels 4q..4q+3] (8 B) and lane l receives the 4 pixels
// of channel l; p already carries the lane's (pixel j, channel quad, 16-channel half) part, o0 / o1 are the byte offsets of the
// operand's first / second group of 4 pixels (immediates).  Addresses must be 8-byte aligned.
__device__ __forceinline__ v8bf vv_tr8(const vv_lds_t p, const int o0, const int o1) {
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + o0));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + o1));
  typedef short v8s __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(v8bf, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int TH, int TW, int NI, int CB, int OB, bool DY16, bool A16>      // DY16 / A16: dy / the layer input hold bf16 elements
__global__ void __launch_bounds__(VV_WG, 1)
wgrad_bf16_kernel(const vv_wgrad_params p, const int NT, const int NCI, const int NCO, const int total, const int nper) {
  constexpr int KW = 4 / (CB * OB);          // waves sharing one (ci-block, co-block) pair: they split the pixels
  constexpr int RL = KW == 4 ? 4 : 8;        // rows a wave walks down per strip
  constexpr int NSEG = KW == 1 ? 2 : 1;      // strips per wave
  constexpr int AHH = TH + 2, AHW = TW + 2;
  constexpr int S = 16;                      // pixel stride: 16 floats = 32 bf16 channels, no pad (transpose reads: see vv_tr8)
  constexpr int APX = NI * AHH * AHW, BPX = NI * TH * TW;
  constexpr int ASZ = APX * S, BSZ = BPX * S;
  static_assert(TH * TW * NI == 256, "16 K steps per tile");
  constexpr int TSZ = CB * ASZ + OB * BSZ;
  constexpr int RSZ = KW > 1 ? CB * OB * 9 * 1024 : 0;      // cross-wave reduction space (one slab per block pair)
  __shared__ float lds[TSZ > RSZ ? TSZ : RSZ];

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int KS = p.ksplit;
  const int NCI2 = NCI / CB, NCO2 = NCO / OB;
  const int ks = w % KS; w /= KS;
  const int cot2 = w % NCO2; w /= NCO2;
  const int cit2 = w % NCI2;
  const int g = w / NCI2;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const int tilesX = W / TW, tilesY = H / TH, tpi = tilesX * tilesY;

  // wave -> (ci block, co block, pixel part)
  const int blk = KW == 1 ? wave : (KW == 2 ? (wave & 1) : 0);
  const int kq = KW == 1 ? 0 : (KW == 2 ? (wave >> 1) : wave);
  const int cb = CB == 2 ? (OB == 2 ? (blk >> 1) : blk) : 0;
  const int ob = OB == 2 ? (blk & 1) : 0;

  const VVSrc sa = vv_make_src(p, g, H, W);
  VVSrc sb;
  sb.p0 = p.dy.ptr + (int64_t)g * p.dy.gstride; sb.cs0 = p.dy.cstride; sb.co0 = p.dy.coff;
  sb.a = sb.b = nullptr; sb.p1 = nullptr; sb.cs1 = sb.co1 = 0; sb.chmap = nullptr; sb.csplit = 0;
  sb.mode = VV_IN_PLAIN; sb.SH = H; sb.SW = W; sb.B = p.B;

  v16f acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // one 32-channel loader per block (a 64-channel block may straddle the skip-concat split; each loader call is uniform)
  // (named objects, not arrays: an array of loaders ends up in scratch memory)
  VVStagerB<NI, AHH, AHW, S, 32> stA0, stA1;
  VVStagerB<NI, TH, TW, S, 32> stB0, stB1;
  stA0.init(sa, -1, tid);
  if constexpr (CB == 2) stA1.init(sa, -1, tid);
  stB0.init(sb, 0, tid);
  if constexpr (OB == 2) stB1.init(sb, 0, tid);
  auto issue = [&](const int pt) __attribute__((always_inline)) {
    const int img0 = (pt / tpi) * NI;
    const int trem = pt % tpi;
    const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;
    if constexpr (A16) {
      stA0.prefetch16x(sa, img0, ty0 - 1, tx0 - 1, (cit2 * CB) * 32, tid, p.CinP);
      if constexpr (CB == 2) stA1.prefetch16x(sa, img0, ty0 - 1, tx0 - 1, (cit2 * CB + 1) * 32, tid, p.CinP);
    } else {
      stA0.prefetch(sa, img0, ty0 - 1, tx0 - 1, (cit2 * CB) * 32, tid, p.CinP);
      if constexpr (CB == 2) stA1.prefetch(sa, img0, ty0 - 1, tx0 - 1, (cit2 * CB + 1) * 32, tid, p.CinP);
    }
    if constexpr (DY16) {
      stB0.prefetch16(sb, img0, ty0, tx0, (cot2 * OB) * 32, tid, p.Cout);
      if constexpr (OB == 2) stB1.prefetch16(sb, img0, ty0, tx0, (cot2 * OB + 1) * 32, tid, p.Cout);
    } else {
      stB0.prefetch(sb, img0, ty0, tx0, (cot2 * OB) * 32, tid, p.Cout);
      if constexpr (OB == 2) stB1.prefetch(sb, img0, ty0, tx0, (cot2 * OB + 1) * 32, tid, p.Cout);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
    if constexpr (A16) {
      stA0.commit16(lds, tid);
      if constexpr (CB == 2) stA1.commit16(lds + ASZ, tid);
    } else {
      stA0.commit_bf16(lds, tid);
      if constexpr (CB == 2) stA1.commit_bf16(lds + ASZ, tid);
    }
    if constexpr (DY16) {
      stB0.commit_raw16(lds + CB * ASZ, tid);
      if constexpr (OB == 2) stB1.commit_raw16(lds + CB * ASZ + BSZ, tid);
    } else {
      stB0.commit_bf16(lds + CB * ASZ, tid);
      if constexpr (OB == 2) stB1.commit_bf16(lds + CB * ASZ + BSZ, tid);
    }
  };

  // transpose-read addressing (vv_tr8): lane 4j+q of a 16-lane group points at pixel j, channels 16*(group&1) + 4q .. +3
  const int trj = (lane >> 2) & 3;
  const int trc = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;            // byte offset inside the pixel's 64 B
  const vv_lds_t xt = vv_lds_ptr(lds + cb * ASZ) + trc;
  const vv_lds_t yt = vv_lds_ptr(lds + CB * ASZ + ob * BSZ) + trc;

  // this workgroup's pixel tiles: ks, ks + KS, ... -- at any moment the workgroups of a launch walk neighbouring tiles
  // (shared halo rows hit in L2, DRAM pages stay open); contiguous ranges per workgroup measured 25 % slower
  issue(ks);
  for (int pt = ks; pt < NT; pt += KS) {
    if (pt != ks) __syncthreads();            // every wave is done reading the previous tile
    commit();
    __syncthreads();
    if (pt + KS < NT) issue(pt + KS);

    if constexpr (TW == 4) {
      // 4x4 level: a lane's 8 pixels are a 2x4 block (rows 2*half, 2*half+1 of image im), a K step is one image.  The nine
      // tap-shifted operands are cut from a 4 x 6 window of the halo tile: 24 reads, 24 packs, no window carried over.
      constexpr int NKS = 16 / KW;
      // lane's 8 pixels of image im: rows 2*half, 2*half+1 (first / second read), columns j (= its position in the read)
      const vv_lds_t xp = xt + ((kq * NKS * AHH + 2 * half) * AHW + trj) * 64;       // halo row 2*half (image row 2*half-1), halo column j
      const vv_lds_t yp = yt + ((kq * NKS * TH + 2 * half) * TW + trj) * 64;
      vv_static_for<0, NKS>([&](auto KK) {
        constexpr int io = KK.value * AHH * AHW * 64, yo = KK.value * TH * TW * 64;
        const v8bf bq = vv_tr8(yp, yo, yo + TW * 64);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const v8bf a0 = vv_tr8(xp, io + (ky * AHW + 0) * 64, io + ((ky + 1) * AHW + 0) * 64);
          const v8bf a1 = vv_tr8(xp, io + (ky * AHW + 1) * 64, io + ((ky + 1) * AHW + 1) * 64);
          const v8bf a2 = vv_tr8(xp, io + (ky * AHW + 2) * 64, io + ((ky + 1) * AHW + 2) * 64);
          acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq, acc[ky * 3 + 0], 0, 0, 0);
          acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq, acc[ky * 3 + 1], 0, 0, 0);
          acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq, acc[ky * 3 + 2], 0, 0, 0);
        }
      });
    } else {
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
      const int strip = KW == 1 ? sg : (KW == 2 ? kq : (kq >> 1));
      const int r0 = KW == 4 ? 4 * (kq & 1) : 0;
      // this lane's 8 pixels of a K step: image im, columns c0 .. c0+7 of tile row rbase + r0 + k
      int im, rbase, c0;
      if constexpr (TW == 32) { im = 0; rbase = 0; c0 = 16 * strip + 8 * half; }
      else if constexpr (TW == 16) { im = 0; rbase = 8 * strip; c0 = 8 * half; }
      else { im = 2 * strip + half; rbase = 0; c0 = 0; }
      const int R = rbase + r0;
      // this lane's pixel of a read: halo row R (image row R-1), halo column c0 + j (image column c0 + j - 1); the second read of an
      // operand is 4 pixels further, tap (ky, kx) and the row walked are immediate offsets
      const vv_lds_t xp = xt + ((im * AHH + R) * AHW + c0 + trj) * 64;
      const vv_lds_t yp = yt + ((im * TH + R) * TW + c0 + trj) * 64;
      v8bf win[3][3];                                                       // [halo row slot][column shift]
      win[0][0] = vv_tr8(xp, (0 * AHW + 0) * 64, (0 * AHW + 4) * 64);
      win[0][1] = vv_tr8(xp, (0 * AHW + 1) * 64, (0 * AHW + 5) * 64);
      win[0][2] = vv_tr8(xp, (0 * AHW + 2) * 64, (0 * AHW + 6) * 64);
      win[1][0] = vv_tr8(xp, (1 * AHW + 0) * 64, (1 * AHW + 4) * 64);
      win[1][1] = vv_tr8(xp, (1 * AHW + 1) * 64, (1 * AHW + 5) * 64);
      win[1][2] = vv_tr8(xp, (1 * AHW + 2) * 64, (1 * AHW + 6) * 64);
      vv_static_for<0, RL>([&](auto KK) {
        constexpr int k = KK.value;
        constexpr int ro = (k + 2) * AHW * 64;
        win[(k + 2) % 3][0] = vv_tr8(xp, ro + 0 * 64, ro + 4 * 64);
        win[(k + 2) % 3][1] = vv_tr8(xp, ro + 1 * 64, ro + 5 * 64);
        win[(k + 2) % 3][2] = vv_tr8(xp, ro + 2 * 64, ro + 6 * 64);
        const v8bf bq = vv_tr8(yp, k * TW * 64, (k * TW + 4) * 64);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(win[(k + ky) % 3][kx], bq, acc[ky * 3 + kx], 0, 0, 0);
      });
    }
    }
  }

  // one slab per (ci block, co block): [tap][ci(32)][co(32)], slab index ((cit*NCO + cot) * KS + ks).  Waves that split the
  // pixels of a block pair are summed through LDS first, in wave order (fixed: bitwise reproducible).
  const int cit = cit2 * CB + cb, cot = cot2 * OB + ob;
  float* out = p.partial + (int64_t)g * p.partial_gstride + ((int64_t)((cit * NCO + cot) * KS + ks)) * (9 * 1024);
  if constexpr (KW == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
        out[t * 1024 + row * 32 + l31] = acc[t][i];
      }
  } else {
    float* red = lds + blk * (9 * 1024);
    __syncthreads();                          // the last tile is no longer read
    for (int wv = 0; wv < KW; ++wv) {
      if (kq == wv) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
            float* q = red + t * 1024 + row * 32 + l31;
            const float v = wv ? *q + acc[t][i] : acc[t][i];
            if (wv == KW - 1) out[t * 1024 + row * 32 + l31] = v; else *q = v;
          }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Transposed convolution (k3, s2, p1, op1):  dW[tap][ci][co] = sum_pixels bf16(act[pixel][ci]) * bf16(dy[2*pixel - 1 + tap][co]).
// Same scheme with the roles swapped: the un-shifted operand is the layer input (8 reads), the tap-shifted one is the output
// gradient, gathered at stride 2 from a (2TH+1) x (2TW+1) halo tile -- the three column taps of a row share 17 reads (9 per
// row on the 4x4 level, where a lane's 8 pixels are a 2x4 block).  Pixel tiles of 128 (8 K steps) keep the 4x halo tile of up to
// 64 output channels in LDS.
template <int TH, int TW, int NI, int CB, int OB, bool DY16, bool A16>
__global__ void __launch_bounds__(VV_WG, 1)
wgradT_bf16_kernel(const vv_wgrad_params p, const int NT, const int NCI, const int NCO, const int total, const int nper) {
  constexpr int KW = 4 / (CB * OB);
  constexpr int NKS = 8 / KW;                // K steps per wave and tile
  constexpr int BHH = 2 * TH + 1, BHW = 2 * TW + 1;
  constexpr int S = 16;
  constexpr int APX = NI * TH * TW, BPX = NI * BHH * BHW;
  constexpr int ASZ = APX * S, BSZ = BPX * S;
  static_assert(APX == 128, "8 K steps per tile");
  constexpr int TSZ = CB * ASZ + OB * BSZ;
  constexpr int RSZ = KW > 1 ? CB * OB * 9 * 1024 : 0;
  __shared__ float lds[TSZ > RSZ ? TSZ : RSZ];

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int KS = p.ksplit;
  const int NCI2 = NCI / CB, NCO2 = NCO / OB;
  const int ks = w % KS; w /= KS;
  const int cot2 = w % NCO2; w /= NCO2;
  const int cit2 = w % NCI2;
  const int g = w / NCI2;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const int tilesX = W / TW, tilesY = H / TH, tpi = tilesX * tilesY;
  const int blk = KW == 1 ? wave : (KW == 2 ? (wave & 1) : 0);
  const int kq = KW == 1 ? 0 : (KW == 2 ? (wave >> 1) : wave);
  const int cb = CB == 2 ? (OB == 2 ? (blk >> 1) : blk) : 0;
  const int ob = OB == 2 ? (blk & 1) : 0;

  const VVSrc sa = vv_make_src(p, g, H, W);
  VVSrc sb;
  sb.p0 = p.dy.ptr + (int64_t)g * p.dy.gstride; sb.cs0 = p.dy.cstride; sb.co0 = p.dy.coff;
  sb.a = sb.b = nullptr; sb.p1 = nullptr; sb.cs1 = sb.co1 = 0; sb.chmap = nullptr; sb.csplit = 0;
  sb.mode = VV_IN_PLAIN; sb.SH = 2 * H; sb.SW = 2 * W; sb.B = p.B;

  v16f acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  VVStagerB<NI, TH, TW, S, 32> stA0, stA1;
  VVStagerB<NI, BHH, BHW, S, 32> stB0, stB1;
  stA0.init(sa, 0, tid);
  if constexpr (CB == 2) stA1.init(sa, 0, tid);
  stB0.init(sb, -1, tid);
  if constexpr (OB == 2) stB1.init(sb, -1, tid);
  auto issue = [&](const int pt) __attribute__((always_inline)) {
    const int img0 = (pt / tpi) * NI;
    const int trem = pt % tpi;
    const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;
    if constexpr (A16) {
      stA0.prefetch16x(sa, img0, ty0, tx0, (cit2 * CB) * 32, tid, p.CinP);
      if constexpr (CB == 2) stA1.prefetch16x(sa, img0, ty0, tx0, (cit2 * CB + 1) * 32, tid, p.CinP);
    } else {
      stA0.prefetch(sa, img0, ty0, tx0, (cit2 * CB) * 32, tid, p.CinP);
      if constexpr (CB == 2) stA1.prefetch(sa, img0, ty0, tx0, (cit2 * CB + 1) * 32, tid, p.CinP);
    }
    if constexpr (DY16) {
      stB0.prefetch16(sb, img0, 2 * ty0 - 1, 2 * tx0 - 1, (cot2 * OB) * 32, tid, p.Cout);
      if constexpr (OB == 2) stB1.prefetch16(sb, img0, 2 * ty0 - 1, 2 * tx0 - 1, (cot2 * OB + 1) * 32, tid, p.Cout);
    } else {
      stB0.prefetch(sb, img0, 2 * ty0 - 1, 2 * tx0 - 1, (cot2 * OB) * 32, tid, p.Cout);
      if constexpr (OB == 2) stB1.prefetch(sb, img0, 2 * ty0 - 1, 2 * tx0 - 1, (cot2 * OB + 1) * 32, tid, p.Cout);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
    if constexpr (A16) {
      stA0.commit16(lds, tid);
      if constexpr (CB == 2) stA1.commit16(lds + ASZ, tid);
    } else {
      stA0.commit_bf16(lds, tid);
      if constexpr (CB == 2) stA1.commit_bf16(lds + ASZ, tid);
    }
    if constexpr (DY16) {
      stB0.commit_raw16(lds + CB * ASZ, tid);
      if constexpr (OB == 2) stB1.commit_raw16(lds + CB * ASZ + BSZ, tid);
    } else {
      stB0.commit_bf16(lds + CB * ASZ, tid);
      if constexpr (OB == 2) stB1.commit_bf16(lds + CB * ASZ + BSZ, tid);
    }
  };

  const int trj = (lane >> 2) & 3;
  const int trc = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const vv_lds_t xt = vv_lds_ptr(lds + cb * ASZ) + trc;
  const vv_lds_t yt = vv_lds_ptr(lds + CB * ASZ + ob * BSZ) + trc;

  issue(ks);
  for (int pt = ks; pt < NT; pt += KS) {
    if (pt != ks) __syncthreads();
    commit();
    __syncthreads();
    if (pt + KS < NT) issue(pt + KS);

    vv_static_for<0, NKS>([&](auto KK) {
      const int k = kq * NKS + KK.value;          // K step of the tile: 16 pixels, 8 per half-wave
      // this lane's 8 pixels: image im, rows r .. r+GR-1, columns c0 .. c0+GC-1 (GR x GC = 1x8, or 2x4 on the 4x4 level); its pixel
      // in a read is column c0 + j (first read) / c0 + 4 + j or the next row (second read)
      constexpr int GR = TW == 4 ? 2 : 1;
      int im, r, c0;
      if constexpr (TW == 16) { im = 0; r = k; c0 = 8 * half; }
      else if constexpr (TW == 8) { im = half; r = k; c0 = 0; }
      else { im = k; r = 2 * half; c0 = 0; }
      const vv_lds_t xp = xt + ((im * TH + r) * TW + c0 + trj) * 64;
      const vv_lds_t yp = yt + ((im * BHH + 2 * r) * BHW + 2 * (c0 + trj)) * 64;   // halo row 2r (image row 2r-1), halo column 2(c0+j)
      constexpr int X1 = GR == 1 ? 4 * 64 : TW * 64;                 // second read of the un-shifted operand
      constexpr int Y1 = GR == 1 ? 8 * 64 : 2 * BHW * 64;            // ... of the stride-2 gathered one
      const v8bf aq = vv_tr8(xp, 0, X1);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const v8bf b0 = vv_tr8(yp, (ky * BHW + 0) * 64, (ky * BHW + 0) * 64 + Y1);
        const v8bf b1 = vv_tr8(yp, (ky * BHW + 1) * 64, (ky * BHW + 1) * 64 + Y1);
        const v8bf b2 = vv_tr8(yp, (ky * BHW + 2) * 64, (ky * BHW + 2) * 64 + Y1);
        acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, b0, acc[ky * 3 + 0], 0, 0, 0);
        acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, b1, acc[ky * 3 + 1], 0, 0, 0);
        acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, b2, acc[ky * 3 + 2], 0, 0, 0);
      }
    });
  }

  const int cit = cit2 * CB + cb, cot = cot2 * OB + ob;
  float* out = p.partial + (int64_t)g * p.partial_gstride + ((int64_t)((cit * NCO + cot) * KS + ks)) * (9 * 1024);
  if constexpr (KW == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
        out[t * 1024 + row * 32 + l31] = acc[t][i];
      }
  } else {
    float* red = lds + blk * (9 * 1024);
    __syncthreads();
    for (int wv = 0; wv < KW; ++wv) {
      if (kq == wv) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
            float* q = red + t * 1024 + row * 32 + l31;
            const float v = wv ? *q + acc[t][i] : acc[t][i];
            if (wv == KW - 1) out[t * 1024 + row * 32 + l31] = v; else *q = v;
          }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same weight gradient for the all-bf16 bank (layer input AND dy stored as bf16: BASELINE config 4), built around the memory
// system instead of around registers.  The register-staged kernel above keeps ONE pixel tile in flight per CU (its loads are
// issued after a tile's commit and must have landed at the next commit), so a 32 -> 32 channel layer -- 36 MFMAs per wave and
// tile -- spends ~2 us of HBM latency per 0.5 us of matrix work: 2.9 TB/s, matrix pipe busy 0.2.  Here tiles go global -> LDS
// directly (buffer_load_dwordx4 ... lds: 16 B per lane, a wave fills 16 pixels x 64 B, no staging registers) into a ring of
// NBUF tile buffers, NBUF-1 tiles ahead of the one being multiplied; counted s_waitcnt vmcnt + raw s_barrier keep the later
// tiles in flight across the barriers (a __syncthreads() would drain them).  The producer's BatchNorm+ReLU is applied in place in
// LDS (ds_read_b128 -> VALU -> ds_write_b128) once a tile has landed; out-of-image halo pixels are zero-filled by the buffer
// bounds check (plain sources) or by that pass (activated sources: relu(b) != 0).  Operands: ds_read_b64_tr_b16 as above.
// Tiles of 256 pixels for 32 x 32 channel blocks, 128 pixels when a workgroup stages 64 input and / or output channels
// (3 x 41-49 KB of the CU's 160 KB).
// global -> LDS, 16 bytes per lane: the wave's 64 items land at lds_addr + lane * 16 (lds_addr wave-uniform, in M0), the source
// offsets are per lane and bounds-checked (out of range: zeros).  Inline asm on purpose: hipcc tracks an LDS-DMA it knows about as a
// pending LDS write and puts s_waitcnt vmcnt(0) in front of the next ds_read -- which would drain the tiles in flight before
// every MFMA phase.  The counter arithmetic is done by hand in the kernel (s_waitcnt vmcnt(N) + raw s_barrier).
__device__ __forceinline__ void vv_glds16(const __amdgpu_buffer_rsrc_t rs, const unsigned lds_addr, const unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
}

// Eight waves: waves 0-3 multiply (the consumers, one per SIMD), waves 4-7 feed them (the producers, one per SIMD): a producer
// wave queues the DMAs of its share of a tile (16-pixel x 64-byte pieces) NBUF-1 tiles ahead, waits for the piece it loaded
// itself (counted vmcnt -- no barrier needed: every lane activates exactly the 16 bytes it transferred), applies the producing
// layer's BatchNorm+ReLU in place, and meets the consumers at ONE barrier per tile.  An elimination run of the four-wave form
// (every wave loading, activating, multiplying in turn) showed the three phases adding up -- 32 -> 32 channels at 32 x 32:
// MFMA phase alone 101 us, + activation 77, + HBM wait 50 = 228 us -- here they overlap on the SIMDs' two instruction streams.
template <int TH, int TW, int NI, int CB, int OB, int NBUF>
__global__ void __launch_bounds__(2 * VV_WG, 1)
wgrad_ring_kernel(const vv_wgrad_params p, const int NT, const int NCI, const int NCO, const int total, const int nper) {
  constexpr int KW = 4 / (CB * OB);                        // consumer waves sharing one (ci-block, co-block) pair: they split the K steps
  constexpr int AHH = TH + 2, AHW = TW + 2;
  constexpr int APX = NI * AHH * AHW, BPX = NI * TH * TW;
  static_assert(BPX == 256 || BPX == 128, "16 or 8 K steps per tile");
  constexpr int NA = (APX + 15) / 16, NB = BPX / 16;       // pieces (16 pixels x 64 B = 1 KB, one DMA instruction) per 32-channel block
  constexpr int ASZB = NA * 1024, BSZB = NB * 1024;        // bytes
  constexpr int TB = CB * ASZB + OB * BSZB;                // one tile buffer
  // a producer wave takes pieces wave, wave + 4, ... of every block: NPA / NPB DMAs per block (the last one may not exist for this
  // wave: it is then an out-of-range DMA into the dump piece behind the ring, so that every wave queues the same number)
  constexpr int NPA = (NA + 3) / 4, NPB = (NB + 3) / 4;
  constexpr int NLA = CB * NPA, NLB = OB * NPB, NLW = NLA + NLB;      // DMAs per producer wave and tile
  constexpr int RSZ = KW > 1 ? CB * OB * 9 * 4096 : 0;
  constexpr int DUMP = NBUF * TB;                          // 1 KB
  constexpr int LDSB = NBUF * TB + 1024 > RSZ ? NBUF * TB + 1024 : RSZ;
  static_assert(LDSB <= 160 * 1024 && NBUF >= 3 && (NBUF - 2) * NLW <= 63, "ring geometry");
  __shared__ __attribute__((aligned(16))) char ldsc[LDSB];          // the ONLY __shared__ object

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int KS = p.ksplit;
  const int NCI2 = NCI / CB, NCO2 = NCO / OB;
  const int ks = w % KS; w /= KS;
  const int cot2 = w % NCO2; w /= NCO2;
  const int cit2 = w % NCI2;
  const int g = w / NCI2;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave8 >= 4;
  const int wave = wave8 & 3;
  const int H = p.H, W = p.W;
  const int tpi = H / TH;                                  // tiles per image (TW == W on every level)
  const vv_lds_t lbase = (vv_lds_t)ldsc;
  const int ntl = (NT - ks + KS - 1) / KS;                 // tiles of this workgroup: pt = ks, ks + KS, ...

  if (producer) {
    // ---- sources.  A block j covers input channels (cit2*CB + j)*32 ..: skip-concat layers take it from src0 (activated) or src1
    const VVSrc sa = vv_make_src(p, g, H, W);
    const char* ptrA[CB];          // channel 0 of block j at pixel 0 (bf16 elements)
    int csA[CB];
    bool actA[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int c0 = (cit2 * CB + j) * 32;
      const bool second = (sa.mode == VV_IN_CAT) && c0 >= sa.csplit;
      csA[j] = second ? sa.cs1 : sa.cs0;
      ptrA[j] = reinterpret_cast<const char*>(second ? sa.p1 : sa.p0) + 2 * (int64_t)((second ? sa.co1 - sa.csplit : sa.co0) + c0);
      actA[j] = (sa.mode == VV_IN_ACT) || (sa.mode == VV_IN_CAT && !second);
    }
    const int csB = p.dy.cstride;
    const char* ptrB = reinterpret_cast<const char*>(p.dy.ptr + (int64_t)g * p.dy.gstride) + 2 * (int64_t)(p.dy.coff + cot2 * OB * 32);
    constexpr unsigned OOB = 0x80000000u;
    // this lane's part of piece i of a tile (i is compile time): pixel (piece*16 + lane/4), 8 channels (lane%4)*8.  Its byte offset
    // from the tile's first halo pixel never changes (the buffer descriptor moves with the tile), only its validity does: halo row 0 /
    // AHH-1 of a tile at the top / bottom of its image, images past the batch.
    const int lq = lane & 3, lp = lane >> 2;
    unsigned voffA[NLA];
    short hyA[NLA], imA[NLA];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int j = i / NPA;
      const int piece = (i % NPA) * 4 + wave;
      const int px = piece * 16 + lp;
      const int hx = px % AHW, t = px / AHW;
      hyA[i] = (short)(t % AHH);
      imA[i] = (short)(t / AHH);
      const bool ok = piece < NA && px < APX && (unsigned)(hx - 1) < (unsigned)W && (cit2 * CB + j) * 32 + lq * 8 < p.CinP;
      voffA[i] = ok ? (unsigned)(((imA[i] * H + hyA[i]) * W + hx) * csA[j] + lq * 8) * 2u : OOB;
    }
    unsigned voffB[NLB];
    short imB[NLB];
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int j = i / NPB;
      const int piece = (i % NPB) * 4 + wave;
      const int px = piece * 16 + lp;
      const int t = px / TW;
      imB[i] = (short)(t / TH);
      voffB[i] = piece < NB ? (unsigned)(((imB[i] * H + t % TH) * W + px % TW) * csB + j * 32 + lq * 8) * 2u : OOB;
    }
    // BatchNorm scale / shift of this lane's 8 channels, packed for v_pk_fma_f32
    v2f sa2[CB][4], sb2[CB][4];
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = (cit2 * CB + j) * 32 + lq * 8 + e;
        const bool on = actA[j] && c < p.CinP;
        float av = sa.a ? sa.a[on ? c : 0] : 0.f, bv = sa.b ? sa.b[on ? c : 0] : 0.f;
        if (!on) { av = 0.f; bv = 0.f; }
        sa2[j][e >> 1][e & 1] = av;
        sb2[j][e >> 1][e & 1] = bv;
      }
    // the kernel's only ordinary loads: have them back before the first DMA is queued behind them
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(sa2[j][e]), "+v"(sb2[j][e]));

    // queue this wave's NLW DMAs of pixel tile pt into ring slot `slot` (live = false: the same number of DMAs, all out of range:
    // no memory traffic, zeros into a slot nobody reads -- the vmcnt arithmetic holds in the pipeline's tail)
    auto issue = [&](const int pt, const int slot, const bool live) __attribute__((always_inline)) {
      const int img0 = (pt / tpi) * NI;
      const int trow = pt % tpi;
      const bool top = trow == 0, bot = trow == tpi - 1;        // (wave-uniform)
      const int nimg = live ? p.B - img0 : 0;                   // images of this tile inside the batch
      const int64_t pix0 = ((int64_t)img0 * H + trow * TH) * W;    // first output pixel of the tile in the [B][H][W] tensor
      const unsigned sl = (unsigned)(size_t)(lbase + slot * TB + wave * 1024);
      const unsigned dump = (unsigned)(size_t)(lbase + DUMP);
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        // descriptor at the tile's first HALO pixel (one row and one column before its first output pixel; for the first tile of the
        // tensor that is in front of the allocation -- those lanes are out of range by construction and never dereferenced)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(ptrA[j] + 2 * ((pix0 - W - 1) * (int64_t)csA[j])), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int ii = 0; ii < NPA; ++ii) {
          const int i = j * NPA + ii;
          const bool exists = ii * 4 + wave < NA;               // (wave-uniform)
          const bool ok = !(top && hyA[i] == 0) && !(bot && hyA[i] == AHH - 1) && imA[i] < nimg;
          vv_glds16(rs, __builtin_amdgcn_readfirstlane(exists ? sl + j * ASZB + ii * 4096 : dump), ok ? voffA[i] : OOB);
        }
      }
      const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ptrB + 2 * (pix0 * (int64_t)csB)), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int i = 0; i < NLB; ++i) {
        const int j = i / NPB;
        const bool exists = (i % NPB) * 4 + wave < NB;
        vv_glds16(rsB, __builtin_amdgcn_readfirstlane(exists ? sl + CB * ASZB + j * BSZB + (i % NPB) * 4096 : dump), imB[i] < nimg ? voffB[i] : OOB);
      }
    };
    // BatchNorm+ReLU of the producing layer, in place, on the 16 bytes this lane transferred of every activated A piece; zero
    // outside the image (plain sources: the bounds check already wrote the zero padding).  2.5 VALU instructions per value:
    // packed fp32 multiply-add, round to bf16 pairs, ReLU as a packed signed 16-bit max with 0 (negative bf16 = negative int16)
    auto activate = [&](const int pt, const int slot) __attribute__((always_inline)) {
      const int img0 = (pt / tpi) * NI;
      const int trow = pt % tpi;
      const bool top = trow == 0, bot = trow == tpi - 1;
      const vv_lds_t sl = lbase + slot * TB + wave * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < NLA; ++i) {
        const int j = i / NPA;
        if (!actA[j] || (i % NPA) * 4 + wave >= NA) continue;
        uint4* q = (uint4*)(sl + j * ASZB + (i % NPA) * 4096);
        const bool ok = voffA[i] != OOB && !(top && hyA[i] == 0) && !(bot && hyA[i] == AHH - 1) && img0 + imA[i] < p.B;
        const uint4 u = *q;
        const unsigned in[4] = {u.x, u.y, u.z, u.w};
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v2f v = {__builtin_bit_cast(float, in[e] << 16), __builtin_bit_cast(float, in[e] & 0xFFFF0000u)};
          v = __builtin_elementwise_fma(sa2[j][e], v, sb2[j][e]);
          const v2bf r = __builtin_convertvector(v, v2bf);
          typedef short v2s __attribute__((ext_vector_type(2)));
          const v2s m = __builtin_elementwise_max(__builtin_bit_cast(v2s, r), (v2s){0, 0});
          o[e] = ok ? __builtin_bit_cast(unsigned, m) : 0u;
        }
        *q = make_uint4(o[0], o[1], o[2], o[3]);
      }
    };

    // prologue: tiles 0 .. NBUF-2 in flight, tile 0 landed + activated
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d) issue(ks + d * KS, d, d < ntl);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NLW) : "memory");
    activate(ks, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // tile 0 is ready
    for (int it = 0; it < ntl; ++it) {
      // the consumers are multiplying tile `it`; the slot of tile it-1 is free: refill it NBUF-1 tiles ahead, then finish tile it+1
      issue(ks + (it + NBUF - 1) * KS, (it + NBUF - 1) % NBUF, it + NBUF - 1 < ntl);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NLW) : "memory");       // this wave's pieces of tile it+1 have landed
      if (it + 1 < ntl) activate(ks + (it + 1) * KS, (it + 1) % NBUF);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // the tail's dummy DMAs write zeros into ring slots: they must have landed before the consumers reuse the ring as reduction space
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (KW > 1)
      for (int wv = 0; wv < KW; ++wv) __builtin_amdgcn_s_barrier();        // the consumers' reduction barriers (below)
    return;
  }

  // ------------------------------------------------------------------------------------------------ consumers
  const int blk = KW == 1 ? wave : (KW == 2 ? (wave & 1) : 0);
  const int kq = KW == 1 ? 0 : (KW == 2 ? (wave >> 1) : wave);
  const int cb = CB == 2 ? (OB == 2 ? (blk >> 1) : blk) : 0;
  const int ob = OB == 2 ? (blk & 1) : 0;
  v16f acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int trj = (lane >> 2) & 3;
  const int trc = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  const int xoff = cb * ASZB + trc, yoff = CB * ASZB + ob * BSZB + trc;

#ifdef VV_RING_PRIO
  __builtin_amdgcn_s_setprio(VV_RING_PRIO);                      // experiment: the consumers' instructions ahead of their SIMD partner's
#endif
  __builtin_amdgcn_s_barrier();                                  // tile 0 is ready
  for (int it = 0; it < ntl; ++it) {
    const int slot = it % NBUF;
    const vv_lds_t xt = lbase + slot * TB + xoff;
    const vv_lds_t yt = lbase + slot * TB + yoff;
    if constexpr (TW == 4) {
      // 4x4 level: a K step is one image; the lane's 8 pixels are rows 2*half, 2*half+1 (first / second read), column j
      constexpr int NKS = NI / KW;
      const vv_lds_t xp = xt + ((kq * NKS * AHH + 2 * half) * AHW + trj) * 64;
      const vv_lds_t yp = yt + ((kq * NKS * TH + 2 * half) * TW + trj) * 64;
      vv_static_for<0, NKS>([&](auto KK) {
        constexpr int io = KK.value * AHH * AHW * 64, yo = KK.value * TH * TW * 64;
        const v8bf bq = vv_tr8(yp, yo, yo + TW * 64);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const v8bf a0 = vv_tr8(xp, io + (ky * AHW + 0) * 64, io + ((ky + 1) * AHW + 0) * 64);
          const v8bf a1 = vv_tr8(xp, io + (ky * AHW + 1) * 64, io + ((ky + 1) * AHW + 1) * 64);
          const v8bf a2 = vv_tr8(xp, io + (ky * AHW + 2) * 64, io + ((ky + 1) * AHW + 2) * 64);
          acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq, acc[ky * 3 + 0], 0, 0, 0);
          acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq, acc[ky * 3 + 1], 0, 0, 0);
          acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq, acc[ky * 3 + 2], 0, 0, 0);
        }
      });
    } else {
      // strips of K steps: a strip walks RLF rows of (TW == 32: one 16-column half | TW == 16: the image | TW == 8: an image pair)
      constexpr int RLF = TW == 32 ? TH : 8;
      constexpr int NSTR = (BPX / 16) / RLF;
      constexpr int SEGS = NSTR >= KW ? NSTR / KW : 1;            // strips per wave
      constexpr int WPS = NSTR >= KW ? 1 : KW / NSTR;             // waves per strip
      constexpr int RL = RLF / WPS;
      static_assert(RL >= 1, "strip geometry");
#pragma unroll
      for (int sg = 0; sg < SEGS; ++sg) {
        const int strip = NSTR >= KW ? kq * SEGS + sg : kq / WPS;
        const int r0 = NSTR >= KW ? 0 : (kq % WPS) * RL;
        int im, rbase, c0;
        if constexpr (TW == 32) { im = 0; rbase = 0; c0 = 16 * strip + 8 * half; }
        else if constexpr (TW == 16) { im = 0; rbase = 8 * strip; c0 = 8 * half; }
        else { im = 2 * strip + half; rbase = 0; c0 = 0; }
        const int R = rbase + r0;
        vv_lds_t xp = xt + ((im * AHH + R) * AHW + c0 + trj) * 64;
        vv_lds_t yp = yt + ((im * TH + R) * TW + c0 + trj) * 64;
        // one address register per operand stream, opaque to the optimiser: every read below is that register plus an immediate
        // (left alone, the slot rotation is strength-reduced into ~30 per-read address registers bumped by VALU adds every tile,
        // which the MFMAs of the same SIMD then wait behind)
        asm volatile("" : "+v"(xp), "+v"(yp));
        // operands one K step ahead of the MFMAs that consume them: a ring of four halo rows x three column shifts, two dy operands
        v8bf win[4][3], bq[2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) win[r][kx] = vv_tr8(xp, (r * AHW + kx) * 64, (r * AHW + kx + 4) * 64);
        bq[0] = vv_tr8(yp, 0, 4 * 64);
        vv_static_for<0, RL>([&](auto KK) {
          constexpr int k = KK.value;
          if constexpr (k + 1 < RL) {
            constexpr int ro = (k + 3) * AHW * 64;
            // (dy first: the next step's first six MFMAs need only it of these eight reads -- their halo rows are older)
            bq[(k + 1) & 1] = vv_tr8(yp, (k + 1) * TW * 64, ((k + 1) * TW + 4) * 64);
            win[(k + 3) % 4][0] = vv_tr8(xp, ro + 0 * 64, ro + 4 * 64);
            win[(k + 3) % 4][1] = vv_tr8(xp, ro + 1 * 64, ro + 5 * 64);
            win[(k + 3) % 4][2] = vv_tr8(xp, ro + 2 * 64, ro + 6 * 64);
          }
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
              acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(win[(k + ky) % 4][kx], bq[k & 1], acc[ky * 3 + kx], 0, 0, 0);
        });
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (every operand read has returned: the slot may be refilled)
    __builtin_amdgcn_s_barrier();                                // tile it+1 is ready; the producers may refill this slot
  }

  __builtin_amdgcn_s_barrier();                                  // the producers' last (dummy) DMAs have landed: the ring is idle
  const int cit = cit2 * CB + cb, cot = cot2 * OB + ob;
  float* out = p.partial + (int64_t)g * p.partial_gstride + ((int64_t)((cit * NCO + cot) * KS + ks)) * (9 * 1024);
  if constexpr (KW == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
        out[t * 1024 + row * 32 + l31] = acc[t][i];
      }
  } else {
    // the KW waves of a block pair are summed through LDS (the idle ring) in wave order: fixed, bitwise reproducible
    float* red = reinterpret_cast<float*>(ldsc) + blk * (9 * 1024);
    for (int wv = 0; wv < KW; ++wv) {
      if (kq == wv) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
            float* q = red + t * 1024 + row * 32 + l31;
            const float v = wv ? *q + acc[t][i] : acc[t][i];
            if (wv == KW - 1) out[t * 1024 + row * 32 + l31] = v; else *q = v;
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// tile geometry of the ring kernel: 256-pixel tiles (16 K steps, four buffers) for one 32 x 32 channel block pair -- its four consumer
// waves split the K steps --, 128-pixel tiles (three buffers) when a workgroup stages 64 input and / or output channels.  The 4x4
// level stays on the register-staged kernel: a tile of 4x4 images is 2.25x halo, and the ring's producers do not keep up.
struct RGeo { int TH, TW, NI; };
inline bool rgeo(int H, int W, bool one, RGeo* t) {
  if (H != W) return false;
  if (H == 32) { *t = {one ? 8 : 4, 32, 1}; return true; }
  if (H == 16) { *t = {one ? 16 : 8, 16, 1}; return true; }
  if (H == 8) { *t = {8, 8, one ? 4 : 2}; return true; }
  return false;
}

struct BGeo { int TH, TW, NI; };
inline bool bgeo(int kind, int H, int W, BGeo* t) {
  if (H != W) return false;
  if (kind == VV_CONV3) {
    if (H == 32) { *t = {8, 32, 1}; return true; }
    if (H == 16) { *t = {16, 16, 1}; return true; }
    if (H == 8) { *t = {8, 8, 4}; return true; }
    if (H == 4) { *t = {4, 4, 16}; return true; }
    return false;
  }
  if (H == 16) { *t = {8, 16, 1}; return true; }      // transposed conv: H x W = its INPUT resolution
  if (H == 8) { *t = {8, 8, 2}; return true; }
  if (H == 4) { *t = {4, 4, 8}; return true; }
  return false;
}

// (ci blocks, co blocks) of 32 channels a workgroup covers: 2 x 2 when the layer has them (the transposed form on the 4x4 level: two
// blocks -- its 4x halo tile of 8 images is the large one)
inline void block_shape(int kind, int TW, int NCI, int NCO, int* cbk, int* obk) {
  int c = NCI % 2 == 0 ? 2 : 1, o = NCO % 2 == 0 ? 2 : 1;
  if (TW == 4 && c * o == 4 && kind != VV_CONV3) o = 1;      // (3x3 at 4x4: 2 x 2 since the 64-byte pixel stride -- 106 KB of LDS, 458 registers, no spills: 145 -> 94 us / 280 -> 243 us)
  *cbk = c; *obk = o;
}

template <int TH, int TW, int NI, int CB, int OB>
int launch_b(const vv_wgrad_params* p, hipStream_t st) {
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH) * (p->W / TW);
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  if (p->ksplit > NT) return VV_ERR_BAD_ARG;
  const int total = p->G * (NCI / CB) * (NCO / OB) * p->ksplit;
  const int nper = (total + 7) / 8;
  if ((p->pad0 & VV_WGRAD_X_BF16) && (p->pad0 & VV_WGRAD_DY_BF16))
    VV_LAUNCH((wgrad_bf16_kernel<TH, TW, NI, CB, OB, true, true>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NCI, NCO, total, nper);
  else if (p->pad0 & VV_WGRAD_DY_BF16)
    VV_LAUNCH((wgrad_bf16_kernel<TH, TW, NI, CB, OB, true, false>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NCI, NCO, total, nper);
  else
    VV_LAUNCH((wgrad_bf16_kernel<TH, TW, NI, CB, OB, false, false>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NCI, NCO, total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

template <int TH, int TW, int NI, int CB, int OB>
int launch_t(const vv_wgrad_params* p, hipStream_t st) {
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH) * (p->W / TW);
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  if (p->ksplit > NT) return VV_ERR_BAD_ARG;
  const int total = p->G * (NCI / CB) * (NCO / OB) * p->ksplit;
  const int nper = (total + 7) / 8;
  if ((p->pad0 & VV_WGRAD_X_BF16) && (p->pad0 & VV_WGRAD_DY_BF16))
    VV_LAUNCH((wgradT_bf16_kernel<TH, TW, NI, CB, OB, true, true>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NCI, NCO, total, nper);
  else if (p->pad0 & VV_WGRAD_DY_BF16)
    VV_LAUNCH((wgradT_bf16_kernel<TH, TW, NI, CB, OB, true, false>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NCI, NCO, total, nper);
  else
    VV_LAUNCH((wgradT_bf16_kernel<TH, TW, NI, CB, OB, false, false>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NCI, NCO, total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

template <int TH, int TW, int NI>
int dispatch_t(const vv_wgrad_params* p, hipStream_t st) {
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  int c, o;
  block_shape(p->kind, TW, NCI, NCO, &c, &o);
  if (c == 2 && o == 2) { if constexpr (TW != 4) return launch_t<TH, TW, NI, 2, 2>(p, st); }
  if (c == 2) return launch_t<TH, TW, NI, 2, 1>(p, st);
  if (o == 2) return launch_t<TH, TW, NI, 1, 2>(p, st);
  return launch_t<TH, TW, NI, 1, 1>(p, st);
}

template <int TH, int TW, int NI>
int dispatch_b(const vv_wgrad_params* p, hipStream_t st) {
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  int c, o;
  block_shape(p->kind, TW, NCI, NCO, &c, &o);
  if (c == 2 && o == 2) return launch_b<TH, TW, NI, 2, 2>(p, st);
  if (c == 2) return launch_b<TH, TW, NI, 2, 1>(p, st);
  if (o == 2) return launch_b<TH, TW, NI, 1, 2>(p, st);
  return launch_b<TH, TW, NI, 1, 1>(p, st);
}

template <int TH, int TW, int NI, int CB, int OB, int NBUF>
int launch_r(const vv_wgrad_params* p, hipStream_t st) {
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH);
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  if (p->ksplit > NT) return VV_ERR_BAD_ARG;
  const int total = p->G * (NCI / CB) * (NCO / OB) * p->ksplit;
  const int nper = (total + 7) / 8;
  VV_LAUNCH((wgrad_ring_kernel<TH, TW, NI, CB, OB, NBUF>), dim3(nper * 8), dim3(2 * VV_WG), 0, st, *p, NT, NCI, NCO, total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

// (ci blocks, co blocks) per workgroup of the ring kernel: 2 x 2 wherever the layer has them, 2 x 1 on the 4x4 level (three buffers
// of a 2 x 2 tile of 8 images with their halos do not fit 160 KB)
inline void ring_block_shape(int H, int NCI, int NCO, int* cbk, int* obk) {
  (void)H;
  *cbk = NCI % 2 == 0 ? 2 : 1;
  *obk = NCO % 2 == 0 ? 2 : 1;
}

// ONE predicate for "vv_wgrad_bf16_plan sizes this launch for the LDS-ring kernel" (geometry and flags alone), used by the plan
// query and by the launch's refusal below
inline bool ring_planned(int kind, int flags, int H, int W, int CinP, int Cout, RGeo* r, int* cbk, int* obk) {
  const int both = VV_WGRAD_X_BF16 | VV_WGRAD_DY_BF16;
  if (kind != VV_CONV3 || (flags & both) != both || Cout % 32 || CinP <= 0 || CinP % 8) return false;
  const int NCI = (CinP + 31) / 32, NCO = Cout / 32;
  ring_block_shape(H, NCI, NCO, cbk, obk);
  return rgeo(H, W, *cbk == 1 && *obk == 1, r);
}

inline bool ring_ok(const vv_wgrad_params* p) {
  const int both = VV_WGRAD_X_BF16 | VV_WGRAD_DY_BF16;
  if (p->kind != VV_CONV3 || (p->pad0 & both) != both || p->H != p->W || p->H < 8) return false;
  if (p->in_mode != VV_IN_PLAIN && p->in_mode != VV_IN_ACT && p->in_mode != VV_IN_CAT) return false;
  if (p->src0.cstride % 8 || p->src0.coff % 8 || p->dy.cstride % 8 || p->dy.coff % 8 || p->CinP % 8) return false;
  if (p->in_mode == VV_IN_CAT && (p->csplit % 32 || p->src1.cstride % 8 || p->src1.coff % 8)) return false;
  return true;
}

int dispatch_r(const vv_wgrad_params* p, hipStream_t st) {
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  int c, o;
  const int H = p->H == p->W ? p->H : 0;
  ring_block_shape(H, NCI, NCO, &c, &o);
#define VV_RING(H_, TH1, NI1, TH2, NI2, TW_)                                                   \
  if (H == H_) {                                                                                \
    if (c == 1 && o == 1) return launch_r<TH1, TW_, NI1, 1, 1, (H_ == 8 ? 3 : 4)>(p, st);        \
    if (c == 2 && o == 1) return launch_r<TH2, TW_, NI2, 2, 1, 3>(p, st);                       \
    if (c == 1 && o == 2) return launch_r<TH2, TW_, NI2, 1, 2, 3>(p, st);                       \
    return launch_r<TH2, TW_, NI2, 2, 2, 3>(p, st);                                             \
  }
  VV_RING(32, 8, 1, 4, 1, 32)
  VV_RING(16, 16, 1, 8, 1, 16)
  VV_RING(8, 8, 4, 8, 2, 8)
#undef VV_RING
  return VV_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int vv_wgrad_bf16_plan(int32_t kind, int32_t B, int32_t H, int32_t W, int32_t CinP, int32_t Cout, int32_t* ntiles,
                                  int32_t* nblocks, int32_t* kw) {
  // kind: bits 0-7 = VV_CONV3 / VV_CONVT_FWD, bits 8.. = the vv_wgrad_params.pad0 flags the launch will carry (the all-bf16
  // 3x3 weight gradient runs the LDS-ring kernel, whose tiles and block shapes differ)
  const int flags = kind >> 8;
  kind &= 0xff;
  {
    RGeo r;
    int cbk, obk;
    if (ring_planned(kind, flags, H, W, CinP, Cout, &r, &cbk, &obk)) {
      const int NCI = (CinP + 31) / 32, NCO = Cout / 32;
      if (ntiles) *ntiles = ((B + r.NI - 1) / r.NI) * (H / r.TH);
      if (nblocks) *nblocks = (NCI / cbk) * (NCO / obk);
      if (kw) *kw = 1;
      return 1;
    }
  }
  BGeo t;
  if (!bgeo(kind, H, W, &t) || Cout % 32 || CinP <= 0) return 0;
  const int NCI = (CinP + 31) / 32, NCO = Cout / 32;
  int cbk, obk;
  block_shape(kind, t.TW, NCI, NCO, &cbk, &obk);
  if (ntiles) *ntiles = ((B + t.NI - 1) / t.NI) * (H / t.TH) * (W / t.TW);
  if (nblocks) *nblocks = (NCI / cbk) * (NCO / obk);
  if (kw) *kw = 1;      // the waves of a workgroup are summed in LDS: one slab per (ci-tile, co-tile) and k-split part
  return 1;
}

extern "C" int vv_wgrad_bf16(const vv_wgrad_params* p, vv_stream stream) {
  if (!p || !p->src0.ptr || !p->dy.ptr || !p->partial) return VV_ERR_BAD_ARG;
  if (p->Cout % 32 || p->ksplit < 1) return VV_ERR_BAD_ARG;
  if (p->in_mode == VV_IN_POOL || p->in_mode == VV_IN_CUBE) return VV_ERR_UNSUPPORTED;   // feed the materialised tensor
  hipStream_t st = (hipStream_t)stream;
  if ((p->pad0 & VV_WGRAD_DY_BF16) && p->dy.coff % 2) return VV_ERR_BAD_ARG;
  if ((p->pad0 & VV_WGRAD_X_BF16) && !(p->pad0 & VV_WGRAD_DY_BF16)) return VV_ERR_UNSUPPORTED;
  if (ring_ok(p)) return dispatch_r(p, st);
  {
    // vv_wgrad_bf16_plan sizes ksplit / the slabs from (kind, flags, geometry) alone: a launch it planned for the LDS-ring kernel
    // must not fall through to the register-staged one (other tiles and block shapes: slabs would be missing) -- refuse instead
    RGeo r;
    int cbk, obk;
    if (ring_planned(p->kind, p->pad0, p->H, p->W, p->CinP, p->Cout, &r, &cbk, &obk))
      return VV_ERR_BAD_ARG;                      // planned as ring, not ring-launchable (alignment / input mode)
  }
  if (p->kind != VV_CONV3) {                   // weight gradient of the transposed conv (H x W = its input resolution)
    switch (p->H == p->W ? p->H : 0) {
      case 16: return dispatch_t<8, 16, 1>(p, st);
      case 8: return dispatch_t<8, 8, 2>(p, st);
      case 4: return dispatch_t<4, 4, 8>(p, st);
    }
    return VV_ERR_UNSUPPORTED;
  }
  switch (p->H == p->W ? p->H : 0) {
    case 32: return dispatch_b<8, 32, 1>(p, st);
    case 16: return dispatch_b<16, 16, 1>(p, st);
    case 8: return dispatch_b<8, 8, 4>(p, st);
    case 4: return dispatch_b<4, 4, 16>(p, st);
  }
  return VV_ERR_UNSUPPORTED;
}
