// Generic NHWC MFMA convolution for the FlowNet2 forward pass (gfx950, v_mfma_f32_32x32x2_f32, exact fp32).
//
//   conv   : nn.Conv2d(k in {1,3,5,7}, stride in {1,2}, padding=(k-1)//2) [+ LeakyReLU(0.1)]   components/misc.py:8-28,42-44
//   deconv : nn.ConvTranspose2d(k4, s2, p1) [+ LeakyReLU(0.1)]                                  components/misc.py:31-39
//            (also the 2->2 channel "upsampled_flow" layers, FlowNetC.py:53-60)
// FlowNet2 is built with_bn=False (flownet2.py:13), so there is no normalisation on this path.
//
// Same structure as the UNet kernel (vv_conv.hip): one workgroup = 8x32 output pixels x 32/64 output channels,
// input halo tile [(TH-1)*s+R][(TW-1)*s+R][CK+4] in LDS (zero padding by predicate, ragged image edges masked),
// packed weight panels [tap][Cin/8][2][CoutP][4] read from global/L2, 4 MFMAs per (A,B) float4 pair.
// The transposed conv runs as 4 output-parity phases with 2x2 taps each.  Channel counts are arbitrary: K is padded
// with zero weights, N is masked in the epilogue; producers write straight into channel slices of the consumer's
// concat buffer (vv_view), so torch.cat never runs.
#include "vv_common.h"

namespace {

// CP > 0: "row-K" form for the few-channel first layers (FlowNetC conv1 7x7 s2 on 3 channels, FlowNetSD conv0 3x3 on 6): the source
// pixels are exactly CP floats apart, so the R pixels under a filter ROW are one contiguous run of R*CP floats -- K walks that run
// ((kx, c) flattened, padded to CK = ceil8(R*CP) with zero weights) and a "tap" is a filter row: 7 x 32 K steps instead of 49 taps
// x 16 zero-padded channels (3.5x fewer MFMAs), 3 x 24 instead of 9 x 16 for the 6-channel layer.
// N16: layers of at most 16 output channels (FlowNetSD / FlowNetFusion: 82 -> 16 3x3 and 162 -> 16 deconv at full / half
// resolution) contract on v_mfma_f32_16x16x4_f32 -- 16 pixels x 16 channels x 4 K per instruction, a wave's 64 pixels as four M
// blocks against one 16-wide N block -- instead of padding N to the 32 of the 32x32x2 tile (half the MFMA cycles; same panel).
// TW = 16: 8 x 16 pixel tiles (one 32-pixel M block per wave) for the H/64 level -- a 7 x 16 map fills 44 % of an 8 x 32 tile.
template <int R, int STRIDE, int DECONV, int NR, int CK, int CP = 0, int N16 = 0, int TW = 32>
__global__ void __launch_bounds__(VV_WG, 2)
conv2d_mfma_kernel(const vv_conv2d_params p, const int tilesX, const int tilesY, const int NN, const int total,
                   const int nper) {
  constexpr int TH = 8;
  static_assert(TW == 32 || (TW == 16 && !N16 && !CP), "tile width");
  constexpr int HH = DECONV ? TH + 2 : (TH - 1) * STRIDE + R;
  constexpr int HW = DECONV ? TW + 2 : (TW - 1) * STRIDE + R;
  constexpr int SP = DECONV ? 1 : STRIDE;
  constexpr int S = CP ? CP : CK + 4, S4 = S / 4;
  constexpr int NCH = CP ? CP : CK;              // channels staged per halo pixel and chunk
  constexpr int MR = TW / 16, TN = NR * 32;
  constexpr int NTAP = DECONV ? 4 : (CP ? R : R * R);
  static_assert(!CP || (!DECONV && CP % 4 == 0 && CK % 8 == 0 && CK >= R * CP && CK < R * CP + 8), "row-K geometry");
  // 3x3 convolutions and the transposed convolution run the software pipeline of the UNet kernel (vv_conv.hip): the
  // activation halo tile AND the chunk's weight panel go global -> registers -> LDS one chunk ahead of the MFMA loop,
  // which reads only LDS, one ds_read_b128 after every group of 4 MFMAs.  5x5 / 7x7 (two layers per sub-network, weight
  // panels of 51 / 100 KB per chunk) keep the direct form: tile staged per chunk, weights from global/L2.
  constexpr bool PIPE = (DECONV || R == 3) && !CP;
  static_assert(!N16 || (PIPE && NR == 1 && CK % 16 == 0), "16-wide N: pipelined form, one N block");
  constexpr int KGC = CK / 8;
  constexpr int A4 = HH * HW * S4;
  constexpr int B4 = PIPE ? NTAP * KGC * 2 * TN : 0;
  constexpr int NBT = PIPE ? (B4 + VV_WG - 1) / VV_WG : 1;
  __shared__ float4 lds4[A4 + B4 + (CP ? 2 : 0)];
  float* lds = reinterpret_cast<float*>(lds4);
  if constexpr (CP != 0) {
    // the K padding of the tile's last pixel reads past the tile: finite zeros there (they meet zero weights)
    if (threadIdx.x < 2) lds4[A4 + B4 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int KS = p.pad0 > 1 ? p.pad0 : 1;        // split-K over input-channel chunks (tiny-M, huge-K layers)
  const int ks = w % KS; w /= KS;
  const int tx = w % tilesX; w /= tilesX;
  const int ty = w % tilesY; w /= tilesY;
  const int nn = w % NN; w /= NN;
  constexpr int NPH = DECONV ? 4 : 1;
  const int ph = w % NPH;
  const int img = w / NPH;
  const int py = ph >> 1, px = ph & 1;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int H = p.H, W = p.W;
  // tile coordinate space: conv = output pixels, deconv = input pixels (each yields one output per phase)
  const int ty0 = ty * TH, tx0 = tx * TW;
  const int pad = (R - 1) / 2;
  const int oy0 = DECONV ? ty0 - 1 : ty0 * STRIDE - pad;
  const int ox0 = DECONV ? tx0 - 1 : tx0 * STRIDE - pad;

  VVSrc s;
  s.p0 = p.src.ptr; s.cs0 = p.src.cstride; s.co0 = p.src.coff; s.a = s.b = nullptr; s.p1 = nullptr; s.cs1 = s.co1 = 0;
  s.chmap = nullptr; s.csplit = 0; s.mode = VV_IN_PLAIN; s.SH = H; s.SW = W; s.B = p.B;

  const int CoutP = p.CoutP, CinP = p.CinP, KQ = CinP >> 3;
  const int co0 = nn * TN;
  const float* __restrict__ wg = p.w;

  int abase[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int pp = wave * (32 * MR) + m * 32 + l31;
    const int r = pp / TW, c = pp % TW;
    abase[m] = ((r * SP) * HW + c * SP) * S4 + half;
  }
  v16f acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

  v4f acc16[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) acc16[mb] = v4f{0.f, 0.f, 0.f, 0.f};

  const int nchunk = CinP / CK;
  const int cbeg = (nchunk * ks / KS) * CK, cend = (nchunk * (ks + 1) / KS) * CK;
  if constexpr (PIPE) {
    const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);
    const v4f* ldsB = reinterpret_cast<const v4f*>(lds4) + A4;
    // tap t of this workgroup: offset of its input pixel inside the halo tile, and its weight-panel index
    int aofft[NTAP], wtap[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if constexpr (DECONV) {
        // oy = 2*iy - 1 + ky: even rows use ky=1 (iy=r) and ky=3 (iy=r-1); odd rows ky=2 (iy=r) and ky=0 (iy=r+1)
        const int ty_ = t >> 1, tx_ = t & 1;
        const int dy = py ? (ty_ ? 1 : 0) : (ty_ ? -1 : 0), ky = py ? (ty_ ? 0 : 2) : (ty_ ? 3 : 1);
        const int dx = px ? (tx_ ? 1 : 0) : (tx_ ? -1 : 0), kx = px ? (tx_ ? 0 : 2) : (tx_ ? 3 : 1);
        aofft[t] = ((1 + dy) * HW + (1 + dx)) * S4;
        wtap[t] = ky * 4 + kx;
      } else {
        aofft[t] = ((t / R) * HW + (t % R)) * S4;
        wtap[t] = t;
      }
    }
    VVStagerB<1, HH, HW, S, CK> stA;
    stA.init(s, ox0, tid);
    unsigned boff[NBT];
    float4 rb[NBT];
#pragma unroll
    for (int k = 0; k < NBT; ++k) {
      const int it = tid + k * VV_WG;
      const int col = it % TN, row = it / TN;            // row = (tap*KGC + kg)*2 + half
      const int hf = row & 1, tk = row >> 1;
      const int kg = tk % KGC, tp = tk / KGC;
      int wt = 0;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) wt = tp == t ? wtap[t] : wt;
      boff[k] = (B4 % VV_WG == 0 || it < B4) ? (unsigned)(((wt * KQ + kg) * 2 + hf) * CoutP + co0 + col) * 16u : 0x80000000u;
    }
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, 0x7FFFFFFF, 0x00020000);
    auto issue = [&](const int c0) {
      stA.prefetch(s, img, oy0, ox0, c0, tid, p.src.cstride);
      const int so = (c0 >> 3) * 2 * CoutP * 16;
#pragma unroll
      for (int k = 0; k < NBT; ++k) {
        const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rsW, boff[k], so, 0);
        rb[k] = make_float4(v.x, v.y, v.z, v.w);
      }
    };
    auto commit = [&]() {
      stA.commit(lds, tid);
#pragma unroll
      for (int k = 0; k < NBT; ++k) {
        const int it = tid + k * VV_WG;
        if (B4 % VV_WG == 0 || it < B4) lds4[A4 + it] = rb[k];
      }
    };
    constexpr int NIT = NTAP * KGC;
    auto rdA = [&](const int it, const int m) -> v4f {
      v4f v = ldsA[abase[m] + aofft[it / KGC] + (it % KGC) * 2];
      // (pinned by the sched_barrier of its MFMA group; an asm "+v"(v) here would force an lgkmcnt(0) wait right behind the read)
      return v;
    };
    auto rdB = [&](const int it, const int n) -> v4f {
      v4f v = ldsB[(it * 2 + half) * TN + n * 32 + l31];
      // (pinned by the sched_barrier of its MFMA group; an asm "+v"(v) here would force an lgkmcnt(0) wait right behind the read)
      return v;
    };
    // ---- 16-wide N (v_mfma_f32_16x16x4_f32): lane = (pixel or channel l15, K quarter q); a K step is 16 channels
    const int q16 = lane >> 4, l15 = lane & 15;
    int ab16[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int pp = wave * 64 + mb * 16 + l15;
      ab16[mb] = (((pp / TW) * SP) * HW + (pp % TW) * SP) * S4 + q16;
    }
    if (cbeg < cend) issue(cbeg);
    for (int c0 = cbeg; c0 < cend; c0 += CK) {
      if (c0 != cbeg) __syncthreads();      // every wave finished reading the previous chunk
      commit();
      __syncthreads();
      if (c0 + CK < cend) issue(c0 + CK);
      if constexpr (N16 != 0) {
        constexpr int KG16 = CK / 16, NIT16 = NTAP * KG16;
        auto rdA16 = [&](const int it, const int mb) -> v4f { return ldsA[ab16[mb] + aofft[it / KG16] + (it % KG16) * 4]; };
        auto rdB16 = [&](const int it) -> v4f {
          return ldsB[(((it / KG16) * KGC + 2 * (it % KG16) + (q16 >> 1)) * 2 + (q16 & 1)) * TN + l15];
        };
        v4f ga[2][4], gb[2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) ga[0][mb] = rdA16(0, mb);
        gb[0] = rdB16(0);
#pragma unroll
        for (int it = 0; it < NIT16; ++it) {
          const int cur = it & 1, nxt = cur ^ 1;
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            acc16[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[cur][mb].x, gb[cur].x, acc16[mb], 0, 0, 0);
            acc16[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[cur][mb].y, gb[cur].y, acc16[mb], 0, 0, 0);
            acc16[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[cur][mb].z, gb[cur].z, acc16[mb], 0, 0, 0);
            acc16[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[cur][mb].w, gb[cur].w, acc16[mb], 0, 0, 0);
            if (it + 1 < NIT16) {              // the next step's five fragments, spread over this step's four MFMA groups
              ga[nxt][mb] = rdA16(it + 1, mb);
              if (mb == 3) gb[nxt] = rdB16(it + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        continue;
      }
      v4f fa[2][MR], fb[2][NR];
#pragma unroll
      for (int m = 0; m < MR; ++m) fa[0][m] = rdA(0, m);
#pragma unroll
      for (int n = 0; n < NR; ++n) fb[0][n] = rdB(0, n);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        int piece = 0;                       // pieces of the next step's fragments: A[0..MR), then B[0..NR)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].x, fb[cur][n].x, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].y, fb[cur][n].y, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].z, fb[cur][n].z, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].w, fb[cur][n].w, acc[m][n], 0, 0, 0);
            if (it + 1 < NIT) {
              const int last = (m == MR - 1 && n == NR - 1);
              do {
                if (piece < MR) fa[nxt][piece] = rdA(it + 1, piece);
                else if (piece < MR + NR) fb[nxt][piece - MR] = rdB(it + 1, piece - MR);
                ++piece;
              } while (last && piece < MR + NR);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    }
  } else {
    // 1x1 / 5x5 / 7x7: the activation tile is pipelined through registers like above; a chunk's weight panel (up to
    // 100 KB for 7x7) does not fit beside it in LDS, so the B fragments come from global/L2 -- two steps ahead of their
    // use (ring of three register sets), while the A fragments of the next step are read from LDS under the MFMAs.
    const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);
    VVStagerB<1, HH, HW, S, NCH> stA;
    stA.init(s, ox0, tid);
    constexpr int NST = NTAP * KGC;
    const int tapstride = KQ * 2 * CoutP * 4;                     // floats between consecutive taps of the packed panel
    auto rdA = [&](const int st, const int m) -> v4f {
      const int t = st / KGC, kg = st % KGC;
      v4f v = ldsA[abase[m] + (CP ? t * HW : (t / R) * HW + (t % R)) * S4 + kg * 2];
      // (pinned by the sched_barrier of its MFMA group; an asm "+v"(v) here would force an lgkmcnt(0) wait right behind the read)
      return v;
    };
    if (cbeg < cend) stA.prefetch(s, img, oy0, ox0, cbeg, tid, p.src.cstride);
    for (int c0 = cbeg; c0 < cend; c0 += CK) {
      if (c0 != cbeg) __syncthreads();
      stA.commit(lds, tid);
      __syncthreads();
      if (c0 + CK < cend) stA.prefetch(s, img, oy0, ox0, c0 + CK, tid, p.src.cstride);
      const float* wbase = wg + ((int64_t)(((c0 >> 3)) * 2 + half) * CoutP + co0 + l31) * 4;
      auto ldB = [&](const int st, const int n) -> float4 {
        const int t = st / KGC, kg = st % KGC;
        return *reinterpret_cast<const float4*>(wbase + (int64_t)t * tapstride + kg * 2 * CoutP * 4 + n * 128);
      };
      v4f fa[2][MR];
      float4 fb[3][NR];
#pragma unroll
      for (int m = 0; m < MR; ++m) fa[0][m] = rdA(0, m);
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        fb[0][n] = ldB(0, n);
        if (NST > 1) fb[1][n] = ldB(1, n);
      }
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int cur = st & 1, nxt = cur ^ 1, bc = st % 3, bn = (st + 2) % 3;
        int piece = 0;                       // next step's A fragments, then the B fragments of step st+2
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].x, fb[bc][n].x, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].y, fb[bc][n].y, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].z, fb[bc][n].z, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m].w, fb[bc][n].w, acc[m][n], 0, 0, 0);
            const int last = (m == MR - 1 && n == NR - 1);
            do {
              if (piece < MR) {
                if (st + 1 < NST) fa[nxt][piece] = rdA(st + 1, piece);
              } else if (piece < MR + NR) {
                if (st + 2 < NST) fb[bn][piece - MR] = ldB(st + 2, piece - MR);
              }
              ++piece;
            } while (last && piece < MR + NR);
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    }
  }

  // ---- epilogue: bias, LeakyReLU, masked NHWC store into the (possibly shared concat) output buffer
  const int OH = DECONV ? 2 * H : (H + 2 * pad - R) / STRIDE + 1;
  const int OW = DECONV ? 2 * W : (W + 2 * pad - R) / STRIDE + 1;
  const int LH = DECONV ? H : OH, LW = DECONV ? W : OW;      // extent of the tile coordinate space
  if constexpr (N16 != 0) {
    // D of the 16x16 tile: lane (channel l15, row quarter q) holds pixels mb*16 + 4q + i, i = 0..3
    const int q16 = lane >> 4, l15 = lane & 15;
    const int co = co0 + l15;
    const bool split = KS > 1;
    float* base = split ? p.out.ptr + (int64_t)ks * p.B * OH * OW * CoutP : p.out.ptr + p.out.coff;
    const int ocs = split ? CoutP : p.out.cstride;
    const bool cok = split ? true : co < p.Cout;
    const float b = (!split && p.bias && cok) ? p.bias[co] : 0.f;
    const float slope = split ? 1.f : p.slope;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pp = wave * 64 + mb * 16 + q16 * 4 + i;
        const int r = ty0 + pp / TW, c = tx0 + pp % TW;
        if (r < LH && c < LW && cok) {
          const int oy = DECONV ? 2 * r + py : r, ox = DECONV ? 2 * c + px : c;
          float v = acc16[mb][i] + b;
          v = v > 0.f ? v : v * slope;
          base[((int64_t)(img * OH + oy) * OW + ox) * ocs + co] = v;
        }
      }
    return;
  }
  if (KS > 1) {
    // raw partial sums -> workspace [ks][B*OH*OW][CoutP]; vv_conv2d_splitk_finish adds them up (+ bias, activation)
    float* ws = p.out.ptr + (int64_t)ks * p.B * OH * OW * CoutP;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
        const int pp = wave * (32 * MR) + m * 32 + row;
        const int r = ty0 + pp / TW, c = tx0 + pp % TW;
        if (r < LH && c < LW) {
          const int oy = DECONV ? 2 * r + py : r, ox = DECONV ? 2 * c + px : c;
          float* o = ws + ((int64_t)(img * OH + oy) * OW + ox) * CoutP + co0 + l31;
#pragma unroll
          for (int n = 0; n < NR; ++n) o[n * 32] = acc[m][n][i];
        }
      }
    return;
  }
  float* __restrict__ outg = p.out.ptr + p.out.coff;
  const int ocs = p.out.cstride;
  const float slope = p.slope;
  float bias[NR];
  bool cok[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int co = co0 + n * 32 + l31;
    cok[n] = co < p.Cout;
    bias[n] = (p.bias && cok[n]) ? p.bias[co] : 0.f;
  }
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
      const int pp = wave * (32 * MR) + m * 32 + row;
      const int r = ty0 + pp / TW, c = tx0 + pp % TW;
      if (r < LH && c < LW) {
        const int oy = DECONV ? 2 * r + py : r, ox = DECONV ? 2 * c + px : c;
        float* o = outg + ((int64_t)(img * OH + oy) * OW + ox) * ocs + co0 + l31;
#pragma unroll
        for (int n = 0; n < NR; ++n)
          if (cok[n]) {
            float v = acc[m][n][i] + bias[n];
            v = v > 0.f ? v : v * slope;
            o[n * 32] = v;
          }
      }
    }
}

__global__ void __launch_bounds__(VV_WG)
splitk_finish_kernel(const float* __restrict__ ws, const int KS, const int64_t M, const int Cout, const int CoutP,
                     const float* __restrict__ bias, const float slope, float* __restrict__ out, const int ocs) {
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= M * Cout) return;
  const int co = (int)(e % Cout);
  const int64_t pix = e / Cout;
  float v = bias ? bias[co] : 0.f;
  for (int k = 0; k < KS; ++k) v += ws[((int64_t)k * M + pix) * CoutP + co];      // fixed order: deterministic
  v = v > 0.f ? v : v * slope;
  out[pix * ocs + co] = v;
}

__global__ void __launch_bounds__(VV_WG)
pack_conv2d_kernel(const float* __restrict__ w, float* __restrict__ packed, const int taps, const int K, const int KP,
                   const int N, const int NP, const int transposed) {
  const int64_t total = (int64_t)taps * KP * NP;
  const int KQ = KP >> 3;
  for (int64_t d = (int64_t)blockIdx.x * VV_WG + threadIdx.x; d < total; d += (int64_t)gridDim.x * VV_WG) {
    const int j = (int)(d & 3);
    int64_t t = d >> 2;
    const int n = (int)(t % NP); t /= NP;
    const int half = (int)(t & 1); t >>= 1;
    const int kq = (int)(t % KQ);
    const int tap = (int)(t / KQ);
    const int k = kq * 8 + half * 4 + j;
    float v = 0.f;
    if (k < K && n < N)
      v = transposed ? w[((int64_t)k * N + n) * taps + tap]       // ConvTranspose2d weight [Cin=k][Cout=n][ky][kx]
                     : w[((int64_t)n * K + k) * taps + tap];      // Conv2d weight [Cout=n][Cin=k][ky][kx]
    packed[d] = v;
  }
}

__global__ void __launch_bounds__(VV_WG)
upsample4_kernel(const int64_t n, const float* __restrict__ src, float* __restrict__ dst, const int C, const int H,
                 const int W, const int bilinear, const float scale) {
  // NCHW, x4: nn.Upsample(scale_factor=4, mode='nearest' (0) | 'bilinear' align_corners=False (1) | align_corners=True (2)),
  // flownet2.py:28,34,43-44; (2) is what the authors' PyTorch 0.3 computed for 'bilinear' 
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= n) return;
  const int OW = 4 * W, OH = 4 * H;
  const int ox = (int)(e % OW);
  const int oy = (int)((e / OW) % OH);
  const int64_t bc = e / ((int64_t)OW * OH);
  const float* p = src + bc * H * W;
  float v;
  if (bilinear) {
    float sy, sx;
    if (bilinear == 2) {      // src = dst * (in - 1) / (out - 1)
      sy = (H > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f) * (float)oy;
      sx = (W > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f) * (float)ox;
    } else {
      sy = fmaxf((oy + 0.5f) * 0.25f - 0.5f, 0.f);
      sx = fmaxf((ox + 0.5f) * 0.25f - 0.5f, 0.f);
    }
    const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    v = hy * (hx * p[y0 * W + x0] + lx * p[y0 * W + x1]) + ly * (hx * p[y1 * W + x0] + lx * p[y1 * W + x1]);
  } else {
    v = p[min(oy >> 2, H - 1) * W + min(ox >> 2, W - 1)];
  }
  dst[e] = v * scale;
}

// ---------------------------------------------------------------------------------------------------------------
// predict_flow: nn.Conv2d(Cin, 2, 3, 1, 1) (components/misc.py:42-44), twelve of them per FlowNet2 forward on inputs of
// up to 1026 channels.  With two output channels a 32-wide MFMA tile wastes 94 % of the matrix core and the layer is a
// bandwidth problem: one thread per output pixel, the input halo tile staged through LDS in 32-channel chunks (read once
// from HBM/L2), weights [tap][Cin/4][2][4] are wave-uniform and arrive through the scalar cache.
__global__ void __launch_bounds__(VV_WG)
conv3x3_n2_kernel(const float* __restrict__ src, const int scs, const int B, const int H, const int W, const int Cin,
                  const float* __restrict__ wq, const int C4P, const float* __restrict__ bias, const float slope,
                  float* __restrict__ out, const int ocs, const int tilesX, const int tilesY) {
  constexpr int TH = 8, TW = 32, CK = 32, S = CK + 4, HH = TH + 2, HW = TW + 2;
  __shared__ float4 lds4[HH * HW * (S / 4)];
  float* lds = reinterpret_cast<float*>(lds4);
  int w = blockIdx.x;
  const int tx = w % tilesX; w /= tilesX;
  const int ty = w % tilesY;
  const int img = w / tilesY;
  const int tid = threadIdx.x, r = tid / TW, c = tid % TW;
  VVSrc s;
  s.p0 = src; s.cs0 = scs; s.co0 = 0; s.a = s.b = nullptr; s.p1 = nullptr; s.cs1 = s.co1 = 0;
  s.chmap = nullptr; s.csplit = 0; s.mode = VV_IN_PLAIN; s.SH = H; s.SW = W; s.B = B;
  float a0 = 0.f, a1 = 0.f;
  const float4* wq4 = reinterpret_cast<const float4*>(wq);
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    if (c0) __syncthreads();
    vv_stage_tile<1, HH, HW, S, CK>(lds, s, img, ty * TH - 1, tx * TW - 1, c0, tid, (Cin + 3) & ~3);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4* pa = lds4 + ((r + t / 3) * HW + (c + t % 3)) * (S / 4);
      const float4* pw = wq4 + ((int64_t)t * C4P + (c0 >> 2)) * 2;
#pragma unroll
      for (int k = 0; k < CK / 4; ++k) {
        const float4 v = pa[k], w0 = pw[2 * k], w1 = pw[2 * k + 1];
        a0 = fmaf(v.w, w0.w, fmaf(v.z, w0.z, fmaf(v.y, w0.y, fmaf(v.x, w0.x, a0))));
        a1 = fmaf(v.w, w1.w, fmaf(v.z, w1.z, fmaf(v.y, w1.y, fmaf(v.x, w1.x, a1))));
      }
    }
  }
  const int oy = ty * TH + r, ox = tx * TW + c;
  if (oy < H && ox < W) {
    float v0 = a0 + (bias ? bias[0] : 0.f), v1 = a1 + (bias ? bias[1] : 0.f);
    v0 = v0 > 0.f ? v0 : v0 * slope;
    v1 = v1 > 0.f ? v1 : v1 * slope;
    float* o = out + ((int64_t)(img * H + oy) * W + ox) * ocs;
    o[0] = v0;
    o[1] = v1;
  }
}

// Same layer when the image is small and the channel count large (the H/8 ... H/64 pyramid levels: 112 .. 28k pixels x up
// to 1026 channels): LPP lanes share one output pixel, each takes every LPP-th group of 4 input channels straight from
// global memory (the 9x re-read of the small map is served by L2), partial sums meet in a shuffle tree.
template <int LPP>
__global__ void __launch_bounds__(VV_WG)
conv3x3_n2_split_kernel(const float* __restrict__ src, const int scs, const int B, const int H, const int W, const int Cin,
                        const float* __restrict__ wq, const int C4P, const float* __restrict__ bias, const float slope,
                        float* __restrict__ out, const int ocs) {
  const int64_t pix = ((int64_t)blockIdx.x * VV_WG + threadIdx.x) / LPP;
  const int sub = threadIdx.x % LPP;
  const int64_t npix = (int64_t)B * H * W;
  const bool live = pix < npix;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), img = (int)(pix / ((int64_t)W * H));
  const int C4 = (Cin + 3) >> 2;            // source pixels hold ceil4(Cin) finite floats
  const float4* wq4 = reinterpret_cast<const float4*>(wq);
  float a0 = 0.f, a1 = 0.f;
  if (live) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
      const float4* pa = reinterpret_cast<const float4*>(src + ((int64_t)(img * H + yy) * W + xx) * scs);
      const float4* pw = wq4 + (int64_t)t * C4P * 2;
      // four channel groups per trip, their 12 loads issued before the first multiply-add: the loop is a chain of L2 round trips
      // (the map is small), one load in flight per lane made it 4x slower
      int k = sub;
      for (; k + 3 * LPP < C4; k += 4 * LPP) {
        float4 v[4], w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = pa[k + u * LPP];
          w0[u] = pw[2 * (k + u * LPP)];
          w1[u] = pw[2 * (k + u * LPP) + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a0 = fmaf(v[u].w, w0[u].w, fmaf(v[u].z, w0[u].z, fmaf(v[u].y, w0[u].y, fmaf(v[u].x, w0[u].x, a0))));
          a1 = fmaf(v[u].w, w1[u].w, fmaf(v[u].z, w1[u].z, fmaf(v[u].y, w1[u].y, fmaf(v[u].x, w1[u].x, a1))));
        }
      }
      for (; k < C4; k += LPP) {
        const float4 v = pa[k], w0 = pw[2 * k], w1 = pw[2 * k + 1];
        a0 = fmaf(v.w, w0.w, fmaf(v.z, w0.z, fmaf(v.y, w0.y, fmaf(v.x, w0.x, a0))));
        a1 = fmaf(v.w, w1.w, fmaf(v.z, w1.z, fmaf(v.y, w1.y, fmaf(v.x, w1.x, a1))));
      }
    }
  }
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o);
    a1 += __shfl_xor(a1, o);
  }
  if (live && sub == 0) {
    float v0 = a0 + (bias ? bias[0] : 0.f), v1 = a1 + (bias ? bias[1] : 0.f);
    float* o = out + pix * ocs;
    o[0] = v0 > 0.f ? v0 : v0 * slope;
    o[1] = v1 > 0.f ? v1 : v1 * slope;
  }
}

// The same layer on 4 x 16 pixel tiles for the 16 / 32-channel heads at full / half resolution: 4 waves x 64 pixels, wave v takes
// channel slice v of the halo tile in LDS (its weights stay wave-uniform: scalar loads), the four slice sums meet in LDS in wave
// order.  Four times the workgroups of the 8 x 32 form and no zero-padded channels: 29 -> 20 us (16 channels, 448 x 1024), 18 -> 11 us
// (32 channels, 224 x 512).  With 64-channel chunks on the quarter-resolution maps it measured slower than the split form below.
template <int CK>
__global__ void __launch_bounds__(VV_WG)
conv3x3_n2_tile_kernel(const float* __restrict__ src, const int scs, const int B, const int H, const int W, const int Cin,
                       const float* __restrict__ wq, const int C4P, const float* __restrict__ bias, const float slope,
                       float* __restrict__ out, const int ocs, const int tilesX, const int tilesY) {
  constexpr int TH = 4, TW = 16, HH = TH + 2, HW = TW + 2, S4 = CK / 4 + 1, Q = CK / 4, QW = Q / 4;
  static_assert(QW >= 1, "a float4 group per wave");
  __shared__ float4 lds4[HH * HW * S4];
  __shared__ float2 part[4][64];
  int w = blockIdx.x;
  const int tx = w % tilesX; w /= tilesX;
  const int ty = w % tilesY;
  const int img = w / tilesY;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane / TW, c = lane % TW;
  const int C4 = (Cin + 3) >> 2;              // source pixels hold ceil4(Cin) finite floats
  const float4* wq4 = reinterpret_cast<const float4*>(wq);
  float a0 = 0.f, a1 = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    if (c0) __syncthreads();
    // (all of a thread's loads in flight before its first LDS write)
    constexpr int NIT = (HH * HW * Q + VV_WG - 1) / VV_WG;
    float4 stg[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * VV_WG;
      const int hp = it / Q, q = it % Q;
      const int y = ty * TH - 1 + hp / HW, x = tx * TW - 1 + hp % HW;
      stg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < HH * HW * Q && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && (c0 >> 2) + q < C4)
        stg[k] = *reinterpret_cast<const float4*>(src + ((int64_t)(img * H + y) * W + x) * scs + c0 + q * 4);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * VV_WG;
      if (it < HH * HW * Q) lds4[(it / Q) * S4 + it % Q] = stg[k];
    }
    __syncthreads();
    const int g0 = (c0 >> 2) + wave * QW;      // this wave's first channel group of the chunk
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4* pa = lds4 + ((r + t / 3) * HW + (c + t % 3)) * S4 + wave * QW;
      const float4* pw = wq4 + ((int64_t)t * C4P + g0) * 2;
#pragma unroll
      for (int k = 0; k < QW; ++k) {
        if (g0 + k < C4P) {                     // (wave-uniform; the panel ends at C4P groups)
          const float4 v = pa[k], w0 = pw[2 * k], w1 = pw[2 * k + 1];
          a0 = fmaf(v.w, w0.w, fmaf(v.z, w0.z, fmaf(v.y, w0.y, fmaf(v.x, w0.x, a0))));
          a1 = fmaf(v.w, w1.w, fmaf(v.z, w1.z, fmaf(v.y, w1.y, fmaf(v.x, w1.x, a1))));
        }
      }
    }
  }
  part[wave][lane] = make_float2(a0, a1);
  __syncthreads();
  const int oy = ty * TH + r, ox = tx * TW + c;
  if (wave == 0 && oy < H && ox < W) {
    float v0 = ((part[0][lane].x + part[1][lane].x) + part[2][lane].x) + part[3][lane].x + (bias ? bias[0] : 0.f);
    float v1 = ((part[0][lane].y + part[1][lane].y) + part[2][lane].y) + part[3][lane].y + (bias ? bias[1] : 0.f);
    float* o = out + ((int64_t)(img * H + oy) * W + ox) * ocs;
    o[0] = v0 > 0.f ? v0 : v0 * slope;
    o[1] = v1 > 0.f ? v1 : v1 * slope;
  }
}

// The small pyramid levels (under 20k pixels, up to 1026 channels) are chains of L2 round trips: the layer is launched as one
// WAVE PER FILTER TAP -- a workgroup of nine waves owns 64 / LPT pixels, wave t multiplies tap t for all of them, LPT lanes share a
// pixel and split its channel groups (all their loads in flight together), a shuffle tree sums the lanes, and the nine tap sums meet
// in LDS in tap order (fixed order: bitwise reproducible).  Nine times the workgroups' worth of loads in flight of the form above.
template <int LPT>
__global__ void __launch_bounds__(576)
conv3x3_n2_tap_kernel(const float* __restrict__ src, const int scs, const int B, const int H, const int W, const int Cin,
                      const float* __restrict__ wq, const int C4P, const float* __restrict__ bias, const float slope,
                      float* __restrict__ out, const int ocs) {
  constexpr int PP = 64 / LPT;                // pixels per workgroup
  __shared__ float2 part[9][PP];
  const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
  const int sub = lane % LPT, pl = lane / LPT;
  const int64_t npix = (int64_t)B * H * W;
  const int64_t pix = (int64_t)blockIdx.x * PP + pl;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), img = (int)(pix / ((int64_t)W * H));
  const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
  const bool live = pix < npix && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
  const int C4 = (Cin + 3) >> 2;
  float a0 = 0.f, a1 = 0.f;
  if (live) {
    const float4* pa = reinterpret_cast<const float4*>(src + ((int64_t)(img * H + yy) * W + xx) * scs);
    const float4* pw = reinterpret_cast<const float4*>(wq) + (int64_t)t * C4P * 2;
    int k = sub;
    for (; k + 3 * LPT < C4; k += 4 * LPT) {
      float4 v[4], w0[4], w1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = pa[k + u * LPT];
        w0[u] = pw[2 * (k + u * LPT)];
        w1[u] = pw[2 * (k + u * LPT) + 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = fmaf(v[u].w, w0[u].w, fmaf(v[u].z, w0[u].z, fmaf(v[u].y, w0[u].y, fmaf(v[u].x, w0[u].x, a0))));
        a1 = fmaf(v[u].w, w1[u].w, fmaf(v[u].z, w1[u].z, fmaf(v[u].y, w1[u].y, fmaf(v[u].x, w1[u].x, a1))));
      }
    }
    for (; k < C4; k += LPT) {
      const float4 v = pa[k], w0 = pw[2 * k], w1 = pw[2 * k + 1];
      a0 = fmaf(v.w, w0.w, fmaf(v.z, w0.z, fmaf(v.y, w0.y, fmaf(v.x, w0.x, a0))));
      a1 = fmaf(v.w, w1.w, fmaf(v.z, w1.z, fmaf(v.y, w1.y, fmaf(v.x, w1.x, a1))));
    }
  }
#pragma unroll
  for (int o = LPT / 2; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o);
    a1 += __shfl_xor(a1, o);
  }
  if (sub == 0) part[t][pl] = make_float2(a0, a1);
  __syncthreads();
  if (threadIdx.x < PP) {
    const int64_t q = (int64_t)blockIdx.x * PP + threadIdx.x;
    if (q < npix) {
      float v0 = part[0][threadIdx.x].x, v1 = part[0][threadIdx.x].y;
#pragma unroll
      for (int u = 1; u < 9; ++u) {
        v0 += part[u][threadIdx.x].x;
        v1 += part[u][threadIdx.x].y;
      }
      v0 += bias ? bias[0] : 0.f;
      v1 += bias ? bias[1] : 0.f;
      float* o = out + q * ocs;
      o[0] = v0 > 0.f ? v0 : v0 * slope;
      o[1] = v1 > 0.f ? v1 : v1 * slope;
    }
  }
}

// upsampled_flow*: nn.ConvTranspose2d(2, 2, 4, 2, 1[, bias]) (FlowNetC.py:53-60, FlowNetS.py:40-47, FlowNetSD.py:45-52,
// FlowNetFusion.py:35-36): 2x2 taps x 2 channels per output -- one thread per output pixel.
__global__ void __launch_bounds__(VV_WG)
deconv4x4_c2_kernel(const float* __restrict__ src, const int scs, const int B, const int H, const int W,
                    const float* __restrict__ wt, const float* __restrict__ bias, const float slope,
                    float* __restrict__ out, const int ocs) {
  const int OH = 2 * H, OW = 2 * W;
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= (int64_t)B * OH * OW) return;
  const int ox = (int)(e % OW), oy = (int)((e / OW) % OH), img = (int)(e / ((int64_t)OW * OH));
  float v0 = bias ? bias[0] : 0.f, v1 = bias ? bias[1] : 0.f;
  // oy = 2*iy - 1 + ky  ->  ky has the parity of oy+1
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ky = ((oy + 1) & 1) + 2 * a, iy = (oy + 1 - ky) >> 1;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kx = ((ox + 1) & 1) + 2 * b, ix = (ox + 1 - kx) >> 1;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float* p = src + ((int64_t)(img * H + iy) * W + ix) * scs;
      const float x0 = p[0], x1 = p[1];
      // weight [Cin][Cout][4][4]
      v0 = fmaf(x0, wt[(0 * 2 + 0) * 16 + ky * 4 + kx], v0);
      v0 = fmaf(x1, wt[(1 * 2 + 0) * 16 + ky * 4 + kx], v0);
      v1 = fmaf(x0, wt[(0 * 2 + 1) * 16 + ky * 4 + kx], v1);
      v1 = fmaf(x1, wt[(1 * 2 + 1) * 16 + ky * 4 + kx], v1);
    }
  }
  float* o = out + e * ocs;
  o[0] = v0 > 0.f ? v0 : v0 * slope;
  o[1] = v1 > 0.f ? v1 : v1 * slope;
}

template <int R, int STRIDE, int DECONV, int CK, int CP = 0>
int launch2d(const vv_conv2d_params* p, hipStream_t st) {
  const int pad = (R - 1) / 2;
  const int LH = DECONV ? p->H : (p->H + 2 * pad - R) / STRIDE + 1;
  const int LW = DECONV ? p->W : (p->W + 2 * pad - R) / STRIDE + 1;
  const bool narrow = LW <= 16 && !CP;          // the H/64 level: 8 x 16 tiles
  const int tilesY = (LH + 7) / 8, tilesX = narrow ? 1 : (LW + 31) / 32;
  const bool wide = p->CoutP % 64 == 0 && p->Cout > 32;
  const int NN = p->CoutP / (wide ? 64 : 32);
  constexpr bool CAN16 = (DECONV || (R == 3 && STRIDE == 1)) && !CP && CK % 16 == 0;
  const int total = p->B * (DECONV ? 4 : 1) * NN * tilesY * tilesX * (p->pad0 > 1 ? p->pad0 : 1);
  const int nper = (total + 7) / 8;
  if (narrow && wide) {
    if constexpr (CP == 0)
      VV_LAUNCH((conv2d_mfma_kernel<R, STRIDE, DECONV, 2, CK, 0, 0, 16>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, tilesX, tilesY, NN,
                total, nper);
  } else if (wide)
    VV_LAUNCH((conv2d_mfma_kernel<R, STRIDE, DECONV, 2, CK, CP>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, tilesX, tilesY, NN,
              total, nper);
  else if (CAN16 && p->Cout <= 16 && p->CoutP == 32) {
    if constexpr (CAN16)
      VV_LAUNCH((conv2d_mfma_kernel<R, STRIDE, DECONV, 1, CK, 0, 1>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, tilesX, tilesY, NN,
                total, nper);
  } else
    VV_LAUNCH((conv2d_mfma_kernel<R, STRIDE, DECONV, 1, CK, CP>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, tilesX, tilesY, NN,
              total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

}  // namespace

extern "C" int vv_conv2d_mfma(const vv_conv2d_params* p, vv_stream stream) {
  if (!p || !p->src.ptr || !p->w || !p->out.ptr) return VV_ERR_BAD_ARG;
  if ((p->kind != 2 && p->CinP % 16) || p->CoutP % 32 || p->src.cstride % 4 || p->src.coff % 4) return VV_ERR_BAD_ARG;
  if ((int64_t)p->B * p->H * p->W * p->src.cstride >= (1ll << 31)) return VV_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (p->kind == 1) return launch2d<4, 2, 1, 16>(p, st);
  if (p->kind == 2) {
    // row-K form (see conv2d_mfma_kernel, CP): pixels exactly cstride floats apart, Cin = CinP = ceil8(R * cstride) flattened
    // (kx, c) indices per filter row, the panel packed with taps = R; split-K does not apply
    if (p->src.coff != 0 || p->pad0 > 1 || p->Cin != p->CinP) return VV_ERR_BAD_ARG;
    if (p->R == 7 && p->stride == 2 && p->src.cstride == 4 && p->CinP == 32) return launch2d<7, 2, 0, 32, 4>(p, st);
    if (p->R == 3 && p->stride == 1 && p->src.cstride == 8 && p->CinP == 24) return launch2d<3, 1, 0, 24, 8>(p, st);
    return VV_ERR_UNSUPPORTED;
  }
  if (p->kind != 0) return VV_ERR_BAD_ARG;
  switch (p->R * 10 + p->stride) {
    case 11: return launch2d<1, 1, 0, 16>(p, st);
    case 31: return launch2d<3, 1, 0, 16>(p, st);
    case 32: return launch2d<3, 2, 0, 8>(p, st);
    case 52: return launch2d<5, 2, 0, 8>(p, st);
    case 72: return launch2d<7, 2, 0, 8>(p, st);
  }
  return VV_ERR_UNSUPPORTED;
}

extern "C" int vv_conv3x3_n2(const float* src, int32_t src_cstride, int32_t B, int32_t H, int32_t W, int32_t Cin,
                             const float* wq, int32_t C4P, const float* bias, float slope, float* out, int32_t out_cstride,
                             int32_t out_coff, vv_stream stream) {
  if (!src || !wq || !out || src_cstride % 4 || C4P * 4 < ((Cin + 31) & ~31)) return VV_ERR_BAD_ARG;
  if ((int64_t)B * H * W * src_cstride >= (1ll << 31)) return VV_ERR_UNSUPPORTED;
  const int64_t npix = (int64_t)B * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (npix >= 20000 && Cin <= 32) {
    // the 16 / 32-channel heads at full / half resolution: 4 x 16 tiles, the four waves split the channels
    const int tilesY = (H + 3) / 4, tilesX = (W + 15) / 16;
    const dim3 grid(B * tilesY * tilesX);
    if (Cin <= 16)
      VV_LAUNCH(conv3x3_n2_tile_kernel<16>, grid, dim3(VV_WG), 0, st, src, src_cstride, B, H, W, Cin, wq, C4P, bias, slope,
                out + out_coff, out_cstride, tilesX, tilesY);
    else
      VV_LAUNCH(conv3x3_n2_tile_kernel<32>, grid, dim3(VV_WG), 0, st, src, src_cstride, B, H, W, Cin, wq, C4P, bias, slope,
                out + out_coff, out_cstride, tilesX, tilesY);
  } else if (npix >= 100000) {              // full / half resolution, more channels: 8 x 32 tiles, one thread per pixel
    const int tilesY = (H + 7) / 8, tilesX = (W + 31) / 32;
    VV_LAUNCH(conv3x3_n2_kernel, dim3(B * tilesY * tilesX), dim3(VV_WG), 0, st, src, src_cstride, B, H, W, Cin, wq, C4P, bias,
              slope, out + out_coff, out_cstride, tilesX, tilesY);
  } else if (npix >= 20000) {               // quarter resolution: 8 lanes per pixel, the 9x re-read served by L2
    VV_LAUNCH(conv3x3_n2_split_kernel<8>, dim3((unsigned)((npix * 8 + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, st, src,
              src_cstride, B, H, W, Cin, wq, C4P, bias, slope, out + out_coff, out_cstride);
  } else if (Cin >= 320) {
    // many channels on a small map: one wave per filter tap (fewer channels leave its lanes idle: the split form below)
    const int lpt = npix >= 4096 ? 16 : (npix >= 1024 ? 32 : 64);
    const dim3 grid((unsigned)((npix * lpt + 63) / 64));
#define VV_N2_TAP(L_)                                                                                                        \
    VV_LAUNCH(conv3x3_n2_tap_kernel<L_>, grid, dim3(576), 0, st, src, src_cstride, B, H, W, Cin, wq, C4P, bias, slope,       \
              out + out_coff, out_cstride)
    if (lpt == 16) VV_N2_TAP(16); else if (lpt == 32) VV_N2_TAP(32); else VV_N2_TAP(64);
#undef VV_N2_TAP
  } else if (npix >= 4096) {                // H/8: 7168 pixels x up to 386 channels -- 32 lanes per pixel fill the chip
    VV_LAUNCH(conv3x3_n2_split_kernel<32>, dim3((unsigned)((npix * 32 + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, st, src,
              src_cstride, B, H, W, Cin, wq, C4P, bias, slope, out + out_coff, out_cstride);
  } else {
    VV_LAUNCH(conv3x3_n2_split_kernel<64>, dim3((unsigned)((npix * 64 + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, st, src,
              src_cstride, B, H, W, Cin, wq, C4P, bias, slope, out + out_coff, out_cstride);
  }
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_deconv4x4_c2(const float* src, int32_t src_cstride, int32_t B, int32_t H, int32_t W, const float* w,
                               const float* bias, float slope, float* out, int32_t out_cstride, int32_t out_coff,
                               vv_stream stream) {
  if (!src || !w || !out) return VV_ERR_BAD_ARG;
  const int64_t n = (int64_t)B * 4 * H * W;
  VV_LAUNCH(deconv4x4_c2_kernel, dim3((unsigned)((n + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, (hipStream_t)stream, src,
            src_cstride, B, H, W, w, bias, slope, out + out_coff, out_cstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_conv2d_splitk_finish(const float* ws, int32_t ksplit, int64_t M, int32_t Cout, int32_t CoutP,
                                       const float* bias, float slope, float* out, int32_t out_cstride, int32_t out_coff,
                                       vv_stream stream) {
  if (!ws || !out || ksplit < 1) return VV_ERR_BAD_ARG;
  const int64_t n = M * Cout;
  VV_LAUNCH(splitk_finish_kernel, dim3((unsigned)((n + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, (hipStream_t)stream, ws, ksplit, M,
            Cout, CoutP, bias, slope, out + out_coff, out_cstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_pack_conv2d(const float* w, float* packed, int32_t taps, int32_t K, int32_t KP, int32_t N, int32_t NP,
                              int32_t transposed, vv_stream stream) {
  if (!w || !packed || KP % 8 || NP % 32) return VV_ERR_BAD_ARG;
  const int64_t total = (int64_t)taps * KP * NP;
  int64_t nb = (total + VV_WG - 1) / VV_WG;
  if (nb > 16384) nb = 16384;
  VV_LAUNCH(pack_conv2d_kernel, dim3((unsigned)nb), dim3(VV_WG), 0, (hipStream_t)stream, w, packed, taps, K, KP, N, NP,
            transposed);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_upsample4(const float* src, float* dst, int32_t BC, int32_t H, int32_t W, int32_t bilinear, float scale,
                            vv_stream stream) {
  if (!src || !dst) return VV_ERR_BAD_ARG;
  const int64_t n = (int64_t)BC * 16 * H * W;
  VV_LAUNCH(upsample4_kernel, dim3((unsigned)((n + VV_WG - 1) / VV_WG)), dim3(VV_WG), 0, (hipStream_t)stream, n, src, dst, BC,
            H, W, bilinear, scale);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
