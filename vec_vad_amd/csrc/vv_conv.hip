// MFMA implicit-GEMM convolutions for the UNet bank (gfx950, v_mfma_f32_32x32x2_f32 -> exact fp32).
//
// One workgroup = 256 threads (4 waves) = one tile of 256 output pixels x TN (32|64) output channels of one UNet.
//   GEMM view:  M = pixels (B*H*W), N = Cout, K = taps * Cin.
//   A operand : input activations, staged global -> registers -> LDS as a halo tile [NI][HH][HW][CK+4] with the
//               producer's BatchNorm+ReLU (+2x2 max-pool, +skip concat, +frame erasure) applied on the way in,
//               read back as ds_read_b128: lane l gets 4 consecutive channels of pixel (l&31), channel group (l>>5).
//   B operand : pre-packed weight panels [tap][Cin/8][2][Cout][4]; the chunk's panel takes the same route
//               global -> registers -> LDS one chunk ahead, so the MFMA loop reads nothing but LDS.
//   One float4 of A and one of B feed 4 MFMAs (k-pairs (j, j+4) of an 8-channel group).
//   Epilogue  : + bias, NHWC store (128 B contiguous per half-wave), per-channel sum / sum^2 partials for BatchNorm.
//
// Replaces: nn.Conv2d(k3,p1) forward (model/unet.py:10,13), its data gradient, nn.ConvTranspose2d(k3,s2,p1,op1)
// forward (model/unet.py:54) and its data gradient; cuDNN calls in the reference.  By default the 3x3 / stride-1 layers run
// through the Winograd kernel in vv_wino.hip (VV_WINOGRAD=0 brings them back here); the transposed convolution always
// runs here -- forward with all four output-parity phases in one workgroup, data gradient as a stride-2 gather.
#include "vv_common.h"
// steps of distance between an LDS fragment read and its MFMAs in the bf16 kernels (measured on BASELINE config 4: 1 -> 41.1 k
// cubes/s, 2 -> 40.5 k, 3 -> 38.6 k: the ring costs registers, and the kernels are bound by LDS bandwidth, not by its latency)
#ifndef VV_PD_BF
#define VV_PD_BF 1
#endif

namespace {

template <int KIND, int TH, int TW>
struct Geo {
  static constexpr int HH = KIND == VV_CONV3 ? TH + 2 : (KIND == VV_CONVT_FWD ? TH + 1 : 2 * TH + 1);
  static constexpr int HW = KIND == VV_CONV3 ? TW + 2 : (KIND == VV_CONVT_FWD ? TW + 1 : 2 * TW + 1);
  static constexpr int SP = KIND == VV_CONVT_DGRAD ? 2 : 1;  // lane pixel stride inside the halo tile
};

// BF = true: mixed-precision variant (BASELINE config 4).  Same tiles, same fp32 tensors in HBM; the weight panel is packed as
// bf16 (vv_pack_weights mode | 4), the activations are rounded to bf16 on their way into LDS (after the deferred BatchNorm+ReLU) and the contraction runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: one ds_read_b128 per operand now
// carries 8 channels of a 16-channel K step (lanes 0-31: k 0..7, lanes 32-63: k 8..15), 1/16 of the matrix-core time of the
// fp32 instruction -- this variant is bound by HBM / staging, not by MFMA.
// S16 (with BF, plain input): src0 holds bf16 elements -- a dy that BatchNorm backward stored as bf16; copied, not converted.
// MR: 32-pixel row blocks per wave (2 = 256-pixel tiles; 1 = 128-pixel tiles, four workgroups per CU: the bf16 kernels on the 8x8 / 4x4
// levels, where a launch has few, long workgroups and is bound by the chunk round trips)
template <int TH, int TW, int NI, int NR, int KIND, int CK, bool BF, int S16, int MR>      // S16: 0 fp32 sources, 1 plain bf16 src0, 2 ALL sources bf16, 3 = 2 + bn_partial, 4 = fp32 stride-2 gather + bn_partial
__global__ void __launch_bounds__(VV_WG, (MR == 1 && BF) ? 4 : ((BF && KIND == VV_CONVT_DGRAD) ? 1 : ((NR == 1 && KIND != VV_CONVT_FWD && !(BF && NI >= 4)) ? 3 : 2)))
conv_mfma_kernel(const vv_conv_params p, const int NT, const int NN, const int total, const int nper) {
  using G_ = Geo<KIND, TH, TW>;
  constexpr int HH = G_::HH, HW = G_::HW, SP = G_::SP;
  static_assert(!BF || CK % 16 == 0, "bf16 K step = 16 channels");
  // LDS pixel stride (floats): S/4 odd -> conflict-free ds_read_b128.  bf16: CK/2 floats of data + 16 B pad
  constexpr int S = BF ? CK / 2 + ((CK / 8) % 2 ? 8 : 4) : CK + 4;
  constexpr int S4 = S / 4;
  static_assert(S4 % 2 == 1, "odd float4 stride");
  static_assert(TH * TW * NI == 128 * MR, "tile pixels");
  constexpr int TN = NR * 32;
  constexpr int KGC = BF ? CK / 16 : CK / 8;   // K groups per chunk: 8 fp32 channels (4 MFMAs) or 16 bf16 channels (1 MFMA)
  constexpr int A4 = NI * HH * HW * S4; // float4 slots of the activation halo tile
  constexpr int BROWS = 9 * KGC * 2;    // weight panel rows of one chunk: [tap][kg][half]
  constexpr int B4 = BROWS * TN;
  constexpr int NBT = (B4 + VV_WG - 1) / VV_WG;
  // bf16 output tile of the epilogue: [128 MR pixels][TN + 8] bf16 (16 B of padding per row), aliases the staging buffers
  constexpr int ORS = TN + 8, OUT4 = (BF && KIND != VV_CONVT_FWD) ? (128 * MR * ORS * 2 + 15) / 16 : 0;
  constexpr int LDS4 = (A4 + B4) > OUT4 ? (A4 + B4) : OUT4;
  __shared__ float4 lds4[LDS4];
  float* lds = reinterpret_cast<float*>(lds4);
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);
  const v4f* ldsB = reinterpret_cast<const v4f*>(lds4) + A4;

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  // the NN output-channel tiles of one pixel tile are neighbours in the work list (same XCD, same time): they stage the same halo
  // tile, which then crosses HBM once instead of NN times (r02 counters on the stride-2 gather at 8x8: 1466 MB moved for 150 MB)
  const int nn = w % NN; w /= NN;
  const int pt = w % NT; w /= NT;
  const int g = w;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int H = p.H, W = p.W;
  const int tilesX = W / TW, tilesY = H / TH;
  const int tpi = tilesX * tilesY;
  const int img0 = (pt / tpi) * NI;
  const int trem = pt % tpi;
  const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;

  // conv-input coordinate space + tile origin in it
  const int SH = KIND == VV_CONVT_DGRAD ? 2 * H : H, SW = KIND == VV_CONVT_DGRAD ? 2 * W : W;
  const int oy0 = KIND == VV_CONV3 ? ty0 - 1 : (KIND == VV_CONVT_FWD ? ty0 : 2 * ty0 - 1);
  const int ox0 = KIND == VV_CONV3 ? tx0 - 1 : (KIND == VV_CONVT_FWD ? tx0 : 2 * tx0 - 1);

  const VVSrc s = vv_make_src(p, g, SH, SW);
  const int Cout = p.Cout, CinP = p.CinP, KQ = CinP >> 3;
  const int co0 = nn * TN;
  const float* __restrict__ wg = p.w + (int64_t)g * p.w_gstride;

  int abase[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int pp = wave * (32 * MR) + m * 32 + l31;
    const int im = pp / (TH * TW), r = (pp / TW) % TH, c = pp % TW;
    abase[m] = ((im * HH + r * SP) * HW + c * SP) * S4 + half;
  }

  // transposed conv: one accumulator set per output parity phase (py, px) -- all 9 taps of a chunk run in ONE workgroup,
  // each tap feeding the phase it belongs to, so the staged tile + weight panel serve 9 taps like in the 3x3 conv
  constexpr int NACC = KIND == VV_CONVT_FWD ? 4 : NR;
  static_assert(KIND != VV_CONVT_FWD || NR == 1, "transposed conv forward: 32-wide N tiles");
  v16f acc[MR][NACC];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  // vv_conv_params.bn_partial (all-bf16 data-gradient launches, 32-wide N tiles): the first pass of the BatchNorm backward that
  // consumes this output rides on the epilogue's store loop.  The thread's z items (the pixels and 8 channels it will store) are
  // requested HERE, ahead of the whole K loop: asked for in the epilogue they put an HBM round trip into every workgroup's
  // critical path (measured: the 32 -> 32 launch at 32x32 204 -> 292 us).
  constexpr bool BNF = BF && NR == 1 && KIND == VV_CONV3 && S16 == 3;      // its own instantiation: as a run-time option it cost
                                                                            // every 32-wide launch of the 32x32 level 20 % (registers, code in the epilogue)
  static_assert(S16 != 3 || BNF, "S16 == 3: the 32-wide 3x3 launch with BatchNorm-backward sums");
  // BNT (round 6, S16 == 4): the fp32 data gradient of a transposed conv IS dA of the conv layer in front of it (its only consumer:
  // layers 7 / 9 / 11), so the first pass of that layer's BatchNorm backward rides on this launch's epilogue the way it rides on the
  // Winograd data gradients (vv_wino.hip): lane = channel, the lane's 16 z values (its 16 output pixels) requested ahead of the K loop.
  // (The all-bf16 form of the same launch -- z as 16-byte items, sums in the LDS store loop like S16 == 3 -- was built and measured on
  // config 4: 138 / 113 / 98 us per launch became 264 / 189 / 146 for 116 us of reduce passes saved; not kept, profiles/README.md.)
  constexpr bool BNT = !BF && KIND == VV_CONVT_DGRAD && NR == 1 && MR == 1 && S16 == 4;
  static_assert(S16 != 4 || BNT, "S16 == 4: the fp32 stride-2 gather (128-pixel tiles, 32-wide N tiles) with BatchNorm-backward sums");
  float zt[BNT ? 16 : 1];
  if constexpr (BNT) {
    const float* zb = p.bn_z + (int64_t)g * p.bn_z_gstride + co0 + l31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
      const int pp = wave * 32 + row;
      const int im = pp / (TH * TW), r = (pp / TW) % TH, c = pp % TW;
      zt[i] = img0 + im < p.B ? zb[((int64_t)((img0 + im) * H + ty0 + r) * W + tx0 + c) * Cout] : 0.f;
    }
  }
  constexpr int BN_QN = NR * 4, BN_NOUT = 128 * MR * BN_QN / VV_WG;
  const bool bnf = BNF && p.bn_partial != nullptr;
  uint4 zq[BNF ? BN_NOUT : 1];
  if constexpr (BNF) {
    if (bnf) {
      const unsigned short* zh = reinterpret_cast<const unsigned short*>(p.bn_z + (int64_t)g * p.bn_z_gstride) + co0 + (tid % BN_QN) * 8;
#pragma unroll
      for (int k = 0; k < BN_NOUT; ++k) {
        const int pp = (tid + k * VV_WG) / BN_QN;
        const int im = pp / (TH * TW), r = (pp / TW) % TH, c = pp % TW;
        zq[k] = img0 + im < p.B ? *reinterpret_cast<const uint4*>(zh + ((int64_t)((img0 + im) * H + ty0 + r) * W + tx0 + c) * Cout)
                                : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }

  // ---- software pipeline over the K chunks: activation tile (BatchNorm+ReLU deferred to commit) and weight panel of
  // chunk c+1 are in flight in registers while chunk c runs on the matrix cores; nothing but LDS is read in the MFMA loop.
  VVStagerB<NI, HH, HW, S, ((BF && S16 >= 2) ? CK / 2 : CK)> stA;      // all-bf16 sources: 16-byte items of 8 channels
  stA.init(s, ox0, tid);            // every tile spans full rows (TW == W): the column origin is tile independent
  unsigned boff[NBT];
  float4 rb[NBT];
  // K groups per tap in the whole panel: fp32 panel [tap][CinP/8][2][Cout][4], bf16 panel [tap][CinP/16][2][Cout][8] -- in both
  // one (tap, group, half) row is Cout x 16 B and IS the LDS row, so the panel chunk is copied as it is
  const int KGT = BF ? (CinP >> 4) : KQ;
#pragma unroll
  for (int k = 0; k < NBT; ++k) {
    const int it = tid + k * VV_WG;
    const int col = it % TN, row = it / TN;            // row = (tap*KGC + kg)*2 + half
    const int hf = row & 1, tk = row >> 1;
    const int kg = tk % KGC, tap = tk / KGC;
    boff[k] = (B4 % VV_WG == 0 || it < B4) ? (unsigned)(((tap * KGT + kg) * 2 + hf) * Cout + co0 + col) * 16u : 0x80000000u;
  }
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, 0x7FFFFFFF, 0x00020000);
  auto issue = [&](const int c0) {
    if constexpr (BF && S16 >= 2) stA.prefetch16w(s, img0, oy0, ox0, c0, tid);
    else if constexpr (BF && S16 == 1) stA.prefetch16(s, img0, oy0, ox0, c0, tid);
    else stA.prefetch(s, img0, oy0, ox0, c0, tid);
    const int so = (BF ? (c0 >> 4) : (c0 >> 3)) * 2 * Cout * 16;
#pragma unroll
    for (int k = 0; k < NBT; ++k) {
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rsW, boff[k], so, 0);
      rb[k] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  auto commit = [&]() {
    if constexpr (BF && S16 >= 2) stA.commit16w(lds, tid);
    else if constexpr (BF && S16 == 1) stA.commit_raw16(lds, tid);
    else if constexpr (BF) stA.commit_bf16(lds, tid);
    else stA.commit(lds, tid);
#pragma unroll
    for (int k = 0; k < NBT; ++k) {
      const int it = tid + k * VV_WG;
      if (B4 % VV_WG == 0 || it < B4) lds4[A4 + it] = rb[k];
    }
  };

  issue(0);
  for (int c0 = 0; c0 < CinP; c0 += CK) {
    if (c0) __syncthreads();            // every wave finished reading the previous chunk
    commit();
    __syncthreads();
    if (c0 + CK < CinP) issue(c0 + CK);

    {
      // 9 taps x CK/8 channel groups, fully unrolled.  The A / B fragments of step it+1 are read from LDS while step it
      // runs: one ds_read_b128 after each group of 4 MFMAs (pinned with sched_barrier), never a block of LDS issue
      // slots in front of the matrix pipe and never a read that is waited on right after it was issued.
      constexpr int NIT = 9 * KGC;
      // Fragment ring: step it + PD is read from LDS while step it runs (PD = 1: two register sets).
      constexpr int PDW = VV_PD_BF;
      constexpr int PD = BF ? (NIT > PDW ? PDW : NIT - 1) : 1, RING = PD + 1;
      v4f fa[RING][MR], fb[RING][NR];
      auto rdA = [&](const int it, const int m) -> v4f {
        const int tap = it / KGC, kg = it % KGC;
        // transposed conv: oy = 2*iy - 1 + ky  ->  ky = 1: iy = r (even rows), ky = 2: iy = r, ky = 0: iy = r + 1 (odd rows)
        const int dy = KIND == VV_CONVT_FWD ? (tap / 3 == 0) : tap / 3, dx = KIND == VV_CONVT_FWD ? (tap % 3 == 0) : tap % 3;
        return ldsA[abase[m] + (dy * HW + dx) * S4 + kg * 2];
      };
      auto rdB = [&](const int it, const int n) -> v4f {
        const int tap = it / KGC, kg = it % KGC;
        return ldsB[((tap * KGC + kg) * 2 + half) * TN + n * 32 + l31];
      };
      // (the reads are pinned between the MFMA groups by sched_barrier alone: an `asm volatile("" : "+v"(v))` on the loaded value
      //  also pins them but makes the compiler wait for the data right there -- one exposed LDS round trip per read)
#pragma unroll
      for (int s = 0; s < PD; ++s) {
#pragma unroll
        for (int m = 0; m < MR; ++m) fa[s][m] = rdA(s, m);
#pragma unroll
        for (int n = 0; n < NR; ++n) fb[s][n] = rdB(s, n);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int cur = it % RING, nxt = (it + PD) % RING;
        int piece = 0;                       // pieces of step it + PD's fragments: A[0..MR), then B[0..NR)
        const int tapc = it / KGC;
        const int phc = ((tapc / 3 != 1) ? 2 : 0) + ((tapc % 3 != 1) ? 1 : 0);   // output phase of this tap (transposed conv)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            const int an = KIND == VV_CONVT_FWD ? phc : n;
            if constexpr (BF) {
              acc[m][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fa[cur][m]),
                                                                   __builtin_bit_cast(v8bf, fb[cur][n]), acc[m][an], 0, 0, 0);
            } else {
              acc[m][an] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].x, fb[cur][n].x, acc[m][an], 0, 0, 0);
              acc[m][an] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].y, fb[cur][n].y, acc[m][an], 0, 0, 0);
              acc[m][an] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].z, fb[cur][n].z, acc[m][an], 0, 0, 0);
              acc[m][an] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].w, fb[cur][n].w, acc[m][an], 0, 0, 0);
            }
            if (it + PD < NIT) {
              const int last = (m == MR - 1 && n == NR - 1);
              // spread MR+NR reads over MR*NR MFMA groups (the last group takes whatever is left)
              do {
                if (piece < MR) fa[nxt][piece] = rdA(it + PD, piece);
                else if (piece < MR + NR) fb[nxt][piece - MR] = rdB(it + PD, piece - MR);
                ++piece;
              } while (last && piece < MR + NR);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    }
  }

  // ---- epilogue: bias, store, BatchNorm partial statistics
  const int OH = KIND == VV_CONVT_FWD ? 2 * H : H, OW = KIND == VV_CONVT_FWD ? 2 * W : W;
  float* __restrict__ outg = p.out.ptr + (int64_t)g * p.out.gstride + p.out.coff;
  const int ocs = p.out.cstride;
  // VV_CONV_OUT_BF16 (bf16 kernels, data gradients): the output is stored as bf16 (same element indexing, gstride in floats)
  // and the per-tile column sums are those of the stored (rounded) values
  const bool o16 = BF && (p.pad0 & VV_CONV_OUT_BF16);
  __bf16* __restrict__ outh = reinterpret_cast<__bf16*>(p.out.ptr + (int64_t)g * p.out.gstride) + p.out.coff;
  float bias[NR], s1[NR], s2[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    bias[n] = p.bias ? p.bias[(int64_t)g * p.bias_gstride + co0 + n * 32 + l31] : 0.f;
    s1[n] = 0.f; s2[n] = 0.f;
  }
  // VV_CONV_RELU: clamp at 0, else at -inf = a no-op that leaves every value bit-identical (the flag test hoisted out of the
  // store loops: tested per element it cost the 64-wide bf16 data gradient at 32x32 85 us of 270)
  const float relu_lo = (p.pad0 & VV_CONV_RELU) ? 0.f : -__builtin_inff();
  float bta = 0.f, btb = 0.f, bt1 = 0.f, bt2 = 0.f;      // BNT: scale / shift of the lane's channel, its two sums
  bool tile_out = false;
  if constexpr (BF && KIND != VV_CONVT_FWD) tile_out = o16;
  if (tile_out) {
    if constexpr (BF && KIND != VV_CONVT_FWD) {
      // bf16 outputs: 2-byte stores straight from the accumulator layout (lane = channel) are 128 B per instruction and cost a
      // third of the kernel (elimination run, profiles/README.md).  The tile goes through LDS instead and leaves as 16-byte
      // items (8 channels) per lane: 8x fewer store instructions, whole 64-byte runs per pixel.
      constexpr int QN = TN / 8, NOUT = 128 * MR * QN / VV_WG;
      static_assert((128 * MR * QN) % VV_WG == 0 && VV_WG % QN == 0, "output items per thread, one channel group per thread");
      static_assert(!BNF || (QN == BN_QN && NOUT == BN_NOUT), "z items = output items");
      __syncthreads();                    // every wave is done with the staging buffers
      unsigned short* lo = reinterpret_cast<unsigned short*>(lds4);
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
          const int pp = wave * (32 * MR) + m * 32 + row;
          const bool ok = img0 + pp / (TH * TW) < p.B;
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            float v = acc[m][n][i] + bias[n];
            v = v < relu_lo ? relu_lo : v;      // ReLU of the folded eval path; compare + select keeps a NaN (fmaxf would drop it)
            const __bf16 hv = (__bf16)v;
            lo[pp * ORS + n * 32 + l31] = __builtin_bit_cast(unsigned short, hv);
            v = ok ? (float)hv : 0.f;
            s1[n] += v; s2[n] = fmaf(v, v, s2[n]);
          }
        }
      __syncthreads();
      // second output view (vv_conv_params.out1): this thread's 8 channels belong to one of the two tensors
      __bf16* obase = outh + co0 + (tid % QN) * 8;
      if (p.out1.ptr && co0 + (tid % QN) * 8 >= p.osplit)
        obase = reinterpret_cast<__bf16*>(p.out1.ptr + (int64_t)g * p.out1.gstride) + p.out1.coff + (co0 + (tid % QN) * 8 - p.osplit);
      float bna[8], bnb[8], bnm[8], bni[8], bs1[8], bs2[8];
      if constexpr (BNF) {
        if (bnf) {
          const int64_t o = (int64_t)g * p.bn_gstride + co0 + (tid % QN) * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) { bna[j] = p.bn_a[o + j]; bnb[j] = p.bn_b[o + j]; bnm[j] = p.bn_mean[o + j]; bni[j] = p.bn_invstd[o + j]; bs1[j] = 0.f; bs2[j] = 0.f; }
        }
      }
#pragma unroll
      for (int k = 0; k < NOUT; ++k) {
        const int it = tid + k * VV_WG;
        const int pp = it / QN;
        const int im = pp / (TH * TW), r = (pp / TW) % TH, c = pp % TW;
        const int img = img0 + im;
        if (img < p.B) {
          const uint4 v = *reinterpret_cast<const uint4*>(lo + pp * ORS + (tid % QN) * 8);
          *reinterpret_cast<uint4*>(obase + ((int64_t)(img * OH + ty0 + r) * OW + tx0 + c) * ocs) = v;
          if constexpr (BNF) {
            if (bnf) {
              // bn_bwd16_kernel's pass 0 on the STORED (rounded) gradient: dz = dA [a z + b > 0]; sum dz, sum dz * z here (two
              // packed-fp32 instructions per channel pair besides the compare / select), xhat's shift and scale once per thread:
              // sum dz * xhat = invstd * (sum dz * z - mean * sum dz)
              const unsigned dw[4] = {v.x, v.y, v.z, v.w}, zw[4] = {zq[k].x, zq[k].y, zq[k].z, zq[k].w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                typedef float v2f_ __attribute__((ext_vector_type(2)));
                const v2f_ z2 = {__builtin_bit_cast(float, zw[j] << 16), __builtin_bit_cast(float, zw[j] & 0xFFFF0000u)};
                const v2f_ d2 = {__builtin_bit_cast(float, dw[j] << 16), __builtin_bit_cast(float, dw[j] & 0xFFFF0000u)};
                const v2f_ a2 = {bna[2 * j], bna[2 * j + 1]}, b2 = {bnb[2 * j], bnb[2 * j + 1]};
                const v2f_ on = __builtin_elementwise_fma(a2, z2, b2);
                const v2f_ dj = {on.x > 0.f ? d2.x : 0.f, on.y > 0.f ? d2.y : 0.f};
                v2f_ t1 = {bs1[2 * j], bs1[2 * j + 1]}, t2 = {bs2[2 * j], bs2[2 * j + 1]};
                t1 += dj;
                t2 = __builtin_elementwise_fma(dj, z2, t2);
                bs1[2 * j] = t1.x; bs1[2 * j + 1] = t1.y; bs2[2 * j] = t2.x; bs2[2 * j + 1] = t2.y;
              }
            }
          }
        }
      }
      if constexpr (BNF) {
        if (bnf) {
          // the 64 threads with this channel group: 16 lanes of each wave (lane % QN), then the four waves through LDS
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            static_assert(!BNF || QN == 4, "lanes with the same lane % 4");
            bs2[j] = bni[j] * (bs2[j] - bnm[j] * bs1[j]);
            bs1[j] += vv_dpp_ror<4>(bs1[j]); bs2[j] += vv_dpp_ror<4>(bs2[j]);      // rows of 16 lanes: rotate by 4, by 8
            bs1[j] += vv_dpp_ror<8>(bs1[j]); bs2[j] += vv_dpp_ror<8>(bs2[j]);
            bs1[j] += __shfl_xor(bs1[j], 16); bs2[j] += __shfl_xor(bs2[j], 16);
            bs1[j] += __shfl_xor(bs1[j], 32); bs2[j] += __shfl_xor(bs2[j], 32);
          }
          __syncthreads();                // the output tile has left LDS
          if (lane < QN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { lds[wave * 2 * TN + lane * 8 + j] = bs1[j]; lds[wave * 2 * TN + TN + lane * 8 + j] = bs2[j]; }
          }
          __syncthreads();
          if (tid < 2 * TN) {
            const float t = (lds[tid] + lds[2 * TN + tid]) + (lds[4 * TN + tid] + lds[6 * TN + tid]);
            p.bn_partial[((int64_t)(g * NT + pt) * 2) * Cout + (tid / TN) * Cout + co0 + tid % TN] = t;
          }
        }
      }
    }
  } else {
  if constexpr (BNT) {
    const int64_t o = (int64_t)g * p.bn_gstride + co0 + l31;
    bta = p.bn_a[o]; btb = p.bn_b[o];
  }
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
      const int pp = wave * (32 * MR) + m * 32 + row;
      const int im = pp / (TH * TW), r = (pp / TW) % TH, c = pp % TW;
      const int img = img0 + im;
      if constexpr (BNT) {
        // dz = dA [a z + b > 0]; sum dz and sum dz * z per channel (xhat's shift and scale once per workgroup, below)
        const float d = (img < p.B && fmaf(bta, zt[i], btb) > 0.f) ? acc[m][0][i] + bias[0] : 0.f;
        bt1 += d;
        bt2 = fmaf(d, zt[i], bt2);
      }
      if (img < p.B) {
        if constexpr (KIND == VV_CONVT_FWD) {
          const int64_t e = ((int64_t)(img * OH + 2 * (ty0 + r)) * OW + 2 * (tx0 + c)) * ocs + co0 + l31;
#pragma unroll
          for (int ph = 0; ph < 4; ++ph) {
            const float v = acc[m][ph][i] + bias[0];
            if (o16) outh[e + ((ph >> 1) * OW + (ph & 1)) * ocs] = (__bf16)v;
            else outg[e + ((ph >> 1) * OW + (ph & 1)) * ocs] = v;
          }
        } else {
          const int64_t e = ((int64_t)(img * OH + ty0 + r) * OW + tx0 + c) * ocs + co0 + l31;
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            float v = acc[m][n][i] + bias[n];
            v = v < relu_lo ? relu_lo : v;      // eval mode: BatchNorm folded into the filter, ReLU in the epilogue (NaN-propagating: fmaxf would drop a NaN)
            if (o16) {
              const __bf16 hv = (__bf16)v;
              outh[e + n * 32] = hv;
              v = (float)hv;
            } else {
              outg[e + n * 32] = v;
            }
            s1[n] += v; s2[n] = fmaf(v, v, s2[n]);
          }
        }
      }
    }

  }
  if constexpr (BNT) {
    // the lane pair of a channel, then the four waves through LDS in wave order: one row [sum dz | sum dz xhat] per pixel tile
    // (vv_bn_bwd_apply reads it with VV_BNBWD_PARTIALS_PER_TTILE: rows = this launch's tiles)
    bt1 += __shfl_xor(bt1, 32);
    bt2 += __shfl_xor(bt2, 32);
    __syncthreads();                      // every wave is done with the staging buffers
    if (half == 0) { lds[wave * 64 + l31] = bt1; lds[wave * 64 + 32 + l31] = bt2; }
    __syncthreads();
    if (tid < 32) {
      const float t1 = (lds[tid] + lds[64 + tid]) + (lds[128 + tid] + lds[192 + tid]);
      const float t2 = (lds[32 + tid] + lds[96 + tid]) + (lds[160 + tid] + lds[224 + tid]);
      const int64_t o = (int64_t)g * p.bn_gstride + co0 + tid;
      float* bp = p.bn_partial + ((int64_t)(g * NT + pt) * 2) * Cout + co0 + tid;
      bp[0] = t1;
      bp[Cout] = p.bn_invstd[o] * (t2 - p.bn_mean[o] * t1);
    }
  }
  if (p.stats) {
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      s1[n] += __shfl_xor(s1[n], 32);
      s2[n] += __shfl_xor(s2[n], 32);
      if (half == 0) {
        lds[(wave * NR + n) * 32 + l31] = s1[n];
        lds[4 * TN + (wave * NR + n) * 32 + l31] = s2[n];
      }
    }
    __syncthreads();
    if (tid < TN) {
      const int n = tid >> 5, l = tid & 31;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        t1 += lds[(wv * NR + n) * 32 + l];
        t2 += lds[4 * TN + (wv * NR + n) * 32 + l];
      }
      float* st = p.stats + ((int64_t)(g * NT + pt) * 2) * Cout + co0 + tid;
      st[0] = t1;
      st[Cout] = t2;
    }
  }
}

struct TileGeo { int TH, TW, NI; };
inline bool tile_geo(int H, int W, TileGeo* t) {
  if (H != W) return false;
  if (H == 32) { *t = {8, 32, 1}; return true; }
  if (H == 16) { *t = {16, 16, 1}; return true; }
  if (H == 8) { *t = {8, 8, 4}; return true; }
  if (H == 4) { *t = {4, 4, 16}; return true; }
  return false;
}

template <int TH, int TW, int NI, int NR, int KIND, int CK, bool BF = false, int S16 = 0, int MR = 2>
int launch(const vv_conv_params* p, hipStream_t st) {
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH) * (p->W / TW);
  const int NN = p->Cout / (NR * 32);
  const int total = p->G * NN * NT;
  const int nper = (total + 7) / 8;
  VV_LAUNCH((conv_mfma_kernel<TH, TW, NI, NR, KIND, CK, BF, S16, MR>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NN,
                     total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

template <int KIND, int CK, bool BF, int S16 = 0>
int dispatch(const vv_conv_params* p, hipStream_t st) {
  TileGeo t;
  if (!tile_geo(p->H, p->W, &t)) return VV_ERR_UNSUPPORTED;
  if constexpr (BF && KIND == VV_CONV3) {      // 128-pixel tiles on the 8x8 / 4x4 levels (vv_conv_ntiles2)
    if (p->H == 16) return (p->Cout % 64) == 0 ? launch<8, 16, 1, 2, KIND, 16, BF, S16, 1>(p, st) : launch<8, 16, 1, 1, KIND, 16, BF, S16, 1>(p, st);
    if (p->H == 8) return (p->Cout % 64) == 0 ? launch<8, 8, 2, 2, KIND, 16, BF, S16, 1>(p, st) : launch<8, 8, 2, 1, KIND, 16, BF, S16, 1>(p, st);
    if (p->H == 4) return (p->Cout % 64) == 0 ? launch<4, 4, 8, 2, KIND, 16, BF, S16, 1>(p, st) : launch<4, 4, 8, 1, KIND, 16, BF, S16, 1>(p, st);
  }
  // 64-wide N tiles halve the activation re-staging but also halve the workgroup count; with 2 workgroups per CU
  // (512 slots) a launch needs >= 2 full rounds of them, otherwise 32-wide tiles fill the machine better.
  const int nt = ((p->B + t.NI - 1) / t.NI) * (p->H / t.TH) * (p->W / t.TW);
  // (bf16 stride-2 gather: the 4x halo tile of a 16-channel chunk takes ~400 registers per lane -- one workgroup per CU)
  // (the fp32 stride-2 gather is bound by its halo loads, not by workgroup count: elimination run -35 % without them; 64-wide
  //  tiles from 512 workgroups on: 230 -> 200 us on the 8x8 level, the 4x4 level with 384 stays 32-wide)
  const int64_t wide_min = (!BF && KIND == VV_CONVT_DGRAD) ? 512 : 1024;
  const bool wide = KIND != VV_CONVT_FWD && (p->Cout % 64) == 0 && (int64_t)p->G * nt * (p->Cout / 64) >= wide_min;
  // K chunk: 16 channels (fp32: 8 for the stride-2 gather and the 16-image 4x4 tiles, whose halo tiles are large)
  constexpr int CKD = BF ? (KIND == VV_CONVT_DGRAD ? 16 : CK) : (KIND == VV_CONVT_DGRAD ? 8 : 16);   // bf16: 16 or 32 (template CK)
  constexpr int CK4 = BF ? 16 : 8;
  if constexpr (!BF && KIND != VV_CONV3) {
    // fp32 transposed conv, 128-pixel tiles (one 32-pixel row block per wave, 64 accumulator registers per phase set): measured at
    // B = 256 against the 256-pixel tiles -- forward 170 -> 131 us (4x4 level) and 155 -> 137 (8x8), 147 -> 158 at 16x16 (kept at 256
    // there); stride-2 gather 178 -> 146 / 191 -> 155 / 188 -> 176 on the three levels.  At the per-rank batches of the reference's
    // DataParallel split (train.py:375) they also double the workgroup count of launches that had a few dozen (B = 32: 99 -> 59 us).
    if (p->H == 16 && KIND == VV_CONVT_DGRAD) return launch<8, 16, 1, 1, KIND, CKD, BF, S16, 1>(p, st);
    if (p->H == 8) return launch<8, 8, 2, 1, KIND, CKD, BF, S16, 1>(p, st);
    if (p->H == 4) return launch<4, 4, 8, 1, KIND, CK4, BF, S16, 1>(p, st);
  }
  if constexpr (KIND == VV_CONVT_FWD) {        // four phase accumulators: 32-wide N tiles only
    switch (p->H) {
      case 32: return launch<8, 32, 1, 1, KIND, CKD, BF, S16>(p, st);
      case 16: return launch<16, 16, 1, 1, KIND, CKD, BF, S16>(p, st);
      case 8: return launch<8, 8, 4, 1, KIND, CKD, BF, S16>(p, st);
      case 4: return launch<4, 4, 16, 1, KIND, CK4, BF, S16>(p, st);
    }
  } else {
    switch (p->H) {
      case 32: return wide ? launch<8, 32, 1, 2, KIND, CKD, BF, S16>(p, st) : launch<8, 32, 1, 1, KIND, CKD, BF, S16>(p, st);
      case 16: return wide ? launch<16, 16, 1, 2, KIND, CKD, BF, S16>(p, st) : launch<16, 16, 1, 1, KIND, CKD, BF, S16>(p, st);
      case 8: return wide ? launch<8, 8, 4, 2, KIND, CKD, BF, S16>(p, st) : launch<8, 8, 4, 1, KIND, CKD, BF, S16>(p, st);
      case 4: return wide ? launch<4, 4, 16, 2, KIND, CK4, BF, S16>(p, st) : launch<4, 4, 16, 1, KIND, CK4, BF, S16>(p, st);
    }
  }
  return VV_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int vv_conv_ntiles(int32_t B, int32_t H, int32_t W) {
  TileGeo t;
  if (!tile_geo(H, W, &t)) return -1;
  return ((B + t.NI - 1) / t.NI) * (H / t.TH) * (W / t.TW);
}

extern "C" int vv_convt_dgrad_ntiles(int32_t B, int32_t H, int32_t W, int32_t flags) {
  // pixel tiles of a VV_CONVT_DGRAD launch of vv_conv_mfma (H x W = its OUTPUT = the transposed conv's input resolution): the rows
  // of bn_partial the fp32 launch leaves.  fp32: 128-pixel tiles; bf16 kernels: 256-pixel tiles (no sums there)
  if (H != W || (H != 16 && H != 8 && H != 4)) return -1;
  const int px = (flags & VV_CONV_BF16) ? 256 : 128;
  const int per_img = H * W;
  return per_img >= px ? B * (per_img / px) : (B + px / per_img - 1) / (px / per_img);
}

extern "C" int vv_conv_ntiles2(int32_t B, int32_t H, int32_t W, int32_t kind, int32_t flags) {
  if (vv_gemm16_flags(kind, flags)) return vv_conv_ntiles(B, H, W);          // vv_conv_bf16.hip: 256-pixel tiles everywhere
  if ((flags & VV_CONV_BF16) && kind == VV_CONV3 && H == W && H == 16) return B * 2;
  if ((flags & VV_CONV_BF16) && kind == VV_CONV3 && H == W && (H == 8 || H == 4)) return (B + (H == 8 ? 1 : 7)) / (H == 8 ? 2 : 8);
  return vv_conv_ntiles(B, H, W);
}

extern "C" int vv_conv_mfma(const vv_conv_params* p, vv_stream stream) {
  if (!p || !p->src0.ptr || !p->w || !p->out.ptr) return VV_ERR_BAD_ARG;
  if (p->G <= 0 || p->B <= 0) return VV_ERR_BAD_ARG;
  if (p->Cout % 32) return VV_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const bool bf = (p->pad0 & VV_CONV_BF16) != 0;
  if (bf && p->CinP % 16) return VV_ERR_BAD_ARG;
  if ((p->pad0 & VV_CONV_SRC_BF16) && !(bf && p->in_mode == VV_IN_PLAIN && p->kind != VV_CONVT_FWD && p->src0.coff % 2 == 0)) return VV_ERR_BAD_ARG;
  if ((p->pad0 & VV_CONV_OUT_BF16) && !bf) return VV_ERR_BAD_ARG;
  if ((p->pad0 & VV_CONV_ALLSRC_BF16) && (!bf || p->in_mode == VV_IN_POOL || p->in_mode == VV_IN_CUBE)) return VV_ERR_BAD_ARG;
  if (p->CinP % 8) return VV_ERR_BAD_ARG;
  if (p->bn_partial && !bf && p->kind == VV_CONVT_DGRAD) {
    // fp32 stride-2 gather (the transposed conv's data gradient = dA of the conv layer in front of it): sums per 128-pixel tile,
    // rows = vv_convt_dgrad_ntiles(B, H, W, 0)
    if (p->stats || p->out1.ptr || (p->H != 16 && p->H != 8 && p->H != 4) || p->H != p->W) return VV_ERR_UNSUPPORTED;
    if (!p->bn_z || !p->bn_a || !p->bn_b || !p->bn_mean || !p->bn_invstd) return VV_ERR_BAD_ARG;
  } else if (p->bn_partial) {
    // BatchNorm-backward partial sums in the epilogue: all-bf16 3x3 launches of the kernel in this file with 32-wide N tiles (the
    // bank's 32 -> 32 channel data gradients on the 32x32 level; vv_conv_wino has its own form for the fp32 path)
    if (!(bf && (p->pad0 & VV_CONV_OUT_BF16) && (p->pad0 & VV_CONV_ALLSRC_BF16) && p->kind == VV_CONV3)) return VV_ERR_UNSUPPORTED;
    if (p->H != 32 || p->W != 32 || p->Cout % 64 == 0 || p->stats || p->out1.ptr) return VV_ERR_UNSUPPORTED;
    if (!p->bn_z || !p->bn_a || !p->bn_b || !p->bn_mean || !p->bn_invstd) return VV_ERR_BAD_ARG;
  }
  if (p->out1.ptr) {                                                    // second output view: bf16-output 3x3 launches only
    if (!(bf && (p->pad0 & VV_CONV_OUT_BF16) && p->kind == VV_CONV3)) return VV_ERR_UNSUPPORTED;
    if (p->osplit <= 0 || p->osplit >= p->Cout || p->osplit % 32 || p->out1.cstride != p->out.cstride ||
        p->out1.gstride != p->out.gstride || p->out1.coff % 8)
      return VV_ERR_BAD_ARG;
  }
  const int sm = !bf ? 0 : ((p->pad0 & VV_CONV_ALLSRC_BF16) ? 2 : ((p->pad0 & VV_CONV_SRC_BF16) ? 1 : 0));
  // all-bf16 3x3 launches: the persistent GEMM-shaped kernel on the 16x16 / 8x8 / 4x4 levels; at 32x32 (HBM-bound, 1 - 2 chunks per
  // tile) it measured slower than this file's kernel, which has the same 256-pixel tiles there (vv_conv_ntiles2 is unaffected)
  if (vv_gemm16_flags(p->kind, p->pad0) && p->H <= 16) return vv_conv_gemm16(p, st);
  if (sm == 2 && vv_conv_ring16_ok(p)) return vv_conv_ring16(p, st);
  switch (p->kind) {
    case VV_CONV3:
      if (p->CinP % 16) return VV_ERR_BAD_ARG;
      // (32-channel chunks for the bf16 kernels: measured -25 % with fp32 input on the 32x32 layers (spills), +-0 end to end with
      // bf16 input -- not kept)
      if (sm == 2 && p->bn_partial) return launch<8, 32, 1, 1, VV_CONV3, 16, true, 3>(p, st);      // (validated above: 32x32 level, Cout % 64 != 0)
      if (sm == 2) return dispatch<VV_CONV3, 16, true, 2>(p, st);
      if (sm == 1) return dispatch<VV_CONV3, 16, true, 1>(p, st);
      return bf ? dispatch<VV_CONV3, 16, true>(p, st) : dispatch<VV_CONV3, 16, false>(p, st);
    case VV_CONVT_FWD:
      if (p->CinP % 16) return VV_ERR_BAD_ARG;
      if (sm == 2) return dispatch<VV_CONVT_FWD, 16, true, 2>(p, st);
      return bf ? dispatch<VV_CONVT_FWD, 16, true>(p, st) : dispatch<VV_CONVT_FWD, 16, false>(p, st);
    case VV_CONVT_DGRAD:
      if (bf && p->CinP % 16) return VV_ERR_BAD_ARG;
      if (!bf && p->bn_partial) {          // (validated above) the tiles of dispatch<VV_CONVT_DGRAD, 8, false>, with the sums
        if (p->H == 16) return launch<8, 16, 1, 1, VV_CONVT_DGRAD, 8, false, 4, 1>(p, st);
        if (p->H == 8) return launch<8, 8, 2, 1, VV_CONVT_DGRAD, 8, false, 4, 1>(p, st);
        return launch<4, 4, 8, 1, VV_CONVT_DGRAD, 8, false, 4, 1>(p, st);
      }
      if (sm == 2) return dispatch<VV_CONVT_DGRAD, 8, true, 2>(p, st);
      if (sm == 1) return dispatch<VV_CONVT_DGRAD, 8, true, 1>(p, st);
      return bf ? dispatch<VV_CONVT_DGRAD, 8, true>(p, st) : dispatch<VV_CONVT_DGRAD, 8, false>(p, st);
  }
  return VV_ERR_BAD_ARG;
}
