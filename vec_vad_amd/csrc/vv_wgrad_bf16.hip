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

// (ci blocks, co blocks) of 32 channels a workgroup covers: 2 x 2 when the layer has them; on the 4x4 levels the staged tiles of
// 16 (8) images are large and two blocks are the most that stays in registers
inline void block_shape(int kind, int TW, int NCI, int NCO, int* cbk, int* obk) {
  int c = NCI % 2 == 0 ? 2 : 1, o = NCO % 2 == 0 ? 2 : 1;
  if (TW == 4 && c * o == 4) { if (kind == VV_CONV3) c = 1; else o = 1; }
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
  if (c == 2 && o == 2) { if constexpr (TW != 4) return launch_b<TH, TW, NI, 2, 2>(p, st); }
  if (c == 2) return launch_b<TH, TW, NI, 2, 1>(p, st);
  if (o == 2) return launch_b<TH, TW, NI, 1, 2>(p, st);
  return launch_b<TH, TW, NI, 1, 1>(p, st);
}

}  // namespace

extern "C" int vv_wgrad_bf16_plan(int32_t kind, int32_t B, int32_t H, int32_t W, int32_t CinP, int32_t Cout, int32_t* ntiles,
                                  int32_t* nblocks, int32_t* kw) {
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
