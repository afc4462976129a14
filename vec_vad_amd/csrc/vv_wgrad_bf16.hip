// Weight gradient of the 3x3 / stride 1 / pad 1 convolution with bf16 operands (mixed precision, BASELINE config 4):
//     dW[tap][ci][co] = sum_pixels bf16(act[pixel + tap][ci]) * bf16(dy[pixel][co]),   fp32 accumulation
// on v_mfma_f32_32x32x16_bf16: M = ci, N = co, K = 16 pixels per instruction.  The instruction wants, per lane, 8 consecutive
// K values (= 8 pixels) of ONE channel, while the tensors are NHWC (channels contiguous).  Instead of transposing through
// LDS, the staged tiles stay pixel-major ([pixel][32 channels] bf16, 72 B pixel stride -- the forward kernel's layout, written
// by the same register-staged loader with the producer's BatchNorm+ReLU applied on the way in) and every lane gathers its 8
// pixels as aligned dwords (lane&31 = channel: two neighbouring lanes share a dword = broadcast, a wave reads 64 contiguous bytes
// per pixel, lanes 32-63 eight pixels further = the other banks) and picks its half while packing pixel pairs (v_perm_b32 with a
// per-lane selector).  The three column-shifted operands of one input row share 10 reads; the three row shifts are a sliding
// window over the rows a wave walks down: 10 + 8 LDS reads and 16 packs per 9 MFMAs.
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

template <int TH, int TW, int NI, int CB, int OB, bool DY16, bool A16>      // DY16 / A16: dy / the layer input hold bf16 elements
__global__ void __launch_bounds__(VV_WG, 1)
wgrad_bf16_kernel(const vv_wgrad_params p, const int NT, const int NCI, const int NCO, const int total, const int nper) {
  constexpr int KW = 4 / (CB * OB);          // waves sharing one (ci-block, co-block) pair: they split the pixels
  constexpr int RL = KW == 4 ? 4 : 8;        // rows a wave walks down per strip
  constexpr int NSEG = KW == 1 ? 2 : 1;      // strips per wave
  constexpr int AHH = TH + 2, AHW = TW + 2;
  constexpr int S = 18, SH = 2 * S;          // pixel stride: 18 floats = 36 bf16 (32 channels + 8 B pad)
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

  // operands are gathered as aligned dwords (two neighbouring lanes = two channels share one: a broadcast, no bank conflict)
  // and the lane's half is selected while packing pixel pairs: v_perm_b32 with a per-lane selector
  const unsigned* ldsw = reinterpret_cast<const unsigned*>(lds);
  const unsigned* xt = ldsw + cb * ASZ + (l31 >> 1);
  const unsigned* yt = ldsw + CB * ASZ + ob * BSZ + (l31 >> 1);
  const unsigned sel = (l31 & 1) ? 0x07060302u : 0x05040100u;       // {hi half of b, hi half of a} : {lo half of b, lo half of a}
  auto pk = [&](const unsigned a, const unsigned b) -> unsigned { return __builtin_amdgcn_perm(b, a, sel); };

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
      vv_static_for<0, NKS>([&](auto KK) {
        const int im = kq * NKS + KK.value;
        const int r = 2 * half;
        const unsigned* xp = xt + ((im * AHH + r) * AHW) * S;      // halo row r (image row r-1), halo column 0 (column -1)
        const unsigned* yp = yt + ((im * TH + r) * TW) * S;
        unsigned P[4][3][2];
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) {
          unsigned v[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) v[j] = xp[(rho * AHW + j) * S];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) { P[rho][kx][0] = pk(v[kx], v[kx + 1]); P[rho][kx][1] = pk(v[kx + 2], v[kx + 3]); }
        }
        unsigned d[8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) d[i * 4 + j] = yp[(i * TW + j) * S];
        const v8bf bq = __builtin_bit_cast(v8bf, (v4u){pk(d[0], d[1]), pk(d[2], d[3]), pk(d[4], d[5]), pk(d[6], d[7])});
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const v4u aq = (v4u){P[ky][kx][0], P[ky][kx][1], P[ky + 1][kx][0], P[ky + 1][kx][1]};
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, aq), bq, acc[ky * 3 + kx], 0, 0, 0);
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
      const unsigned* xp = xt + ((im * AHH + R) * AHW + c0) * S;     // halo row R (image row R-1), halo column c0 (column c0-1)
      const unsigned* yp = yt + ((im * TH + R) * TW + c0) * S;
      v4u win[3][3];                                                        // [halo row slot][column shift]
      auto load_row = [&](const int slot, const int hr) __attribute__((always_inline)) {
        unsigned v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = xp[(hr * AHW + j) * S];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          win[slot][dx] = (v4u){pk(v[dx], v[dx + 1]), pk(v[dx + 2], v[dx + 3]), pk(v[dx + 4], v[dx + 5]), pk(v[dx + 6], v[dx + 7])};
      };
      load_row(0, 0);
      load_row(1, 1);
      vv_static_for<0, RL>([&](auto KK) {
        constexpr int k = KK.value;
        load_row((k + 2) % 3, k + 2);
        unsigned d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = yp[(k * TW + j) * S];
        const v8bf bq = __builtin_bit_cast(v8bf, (v4u){pk(d[0], d[1]), pk(d[2], d[3]), pk(d[4], d[5]), pk(d[6], d[7])});
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, win[(k + ky) % 3][kx]), bq,
                                                                       acc[ky * 3 + kx], 0, 0, 0);
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
  constexpr int S = 18;
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

  const unsigned* ldsw = reinterpret_cast<const unsigned*>(lds);
  const unsigned* xt = ldsw + cb * ASZ + (l31 >> 1);
  const unsigned* yt = ldsw + CB * ASZ + ob * BSZ + (l31 >> 1);
  const unsigned sel = (l31 & 1) ? 0x07060302u : 0x05040100u;
  auto pk = [&](const unsigned a, const unsigned b) -> unsigned { return __builtin_amdgcn_perm(b, a, sel); };

  issue(ks);
  for (int pt = ks; pt < NT; pt += KS) {
    if (pt != ks) __syncthreads();
    commit();
    __syncthreads();
    if (pt + KS < NT) issue(pt + KS);

    vv_static_for<0, NKS>([&](auto KK) {
      const int k = kq * NKS + KK.value;          // K step of the tile: 16 pixels, 8 per half-wave
      // this lane's 8 pixels: image im, rows r .. r+GR-1, columns c0 .. c0+GC-1 (GR x GC = 1x8, or 2x4 on the 4x4 level)
      constexpr int GR = TW == 4 ? 2 : 1, GC = 8 / GR;
      int im, r, c0;
      if constexpr (TW == 16) { im = 0; r = k; c0 = 8 * half; }
      else if constexpr (TW == 8) { im = half; r = k; c0 = 0; }
      else { im = k; r = 2 * half; c0 = 0; }
      const unsigned* xp = xt + ((im * TH + r) * TW + c0) * S;
      const unsigned* yp = yt + ((im * BHH + 2 * r) * BHW + 2 * c0) * S;   // halo row 2r (image row 2r-1), halo column 2c0
      unsigned xv[8];
#pragma unroll
      for (int i = 0; i < GR; ++i)
#pragma unroll
        for (int j = 0; j < GC; ++j) xv[i * GC + j] = xp[(i * TW + j) * S];
      const v8bf aq = __builtin_bit_cast(v8bf, (v4u){pk(xv[0], xv[1]), pk(xv[2], xv[3]), pk(xv[4], xv[5]), pk(xv[6], xv[7])});
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        unsigned dv[GR][2 * GC + 1];
#pragma unroll
        for (int i = 0; i < GR; ++i)
#pragma unroll
          for (int j = 0; j < 2 * GC + 1; ++j) dv[i][j] = yp[((2 * i + ky) * BHW + j) * S];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          v4u bq;
          if constexpr (GR == 1) {
            bq = (v4u){pk(dv[0][kx], dv[0][kx + 2]), pk(dv[0][kx + 4], dv[0][kx + 6]), pk(dv[0][kx + 8], dv[0][kx + 10]),
                       pk(dv[0][kx + 12], dv[0][kx + 14])};
          } else {
            bq = (v4u){pk(dv[0][kx], dv[0][kx + 2]), pk(dv[0][kx + 4], dv[0][kx + 6]), pk(dv[GR - 1][kx], dv[GR - 1][kx + 2]),
                       pk(dv[GR - 1][kx + 4], dv[GR - 1][kx + 6])};
          }
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, __builtin_bit_cast(v8bf, bq), acc[ky * 3 + kx], 0, 0, 0);
        }
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
